"""
Test infrastructure: a kernel backend for the Solver classes that runs on torch
CPU tensors and evaluates every launch with the CPU oracle.  It lets the
decomposition / halo-exchange / reduction logic of sailfish_amd.solvers run
under multi-process gloo tests without a GPU.  It is injected through the
solvers' ``backend=`` keyword; the package itself never imports it.

The fused entry points have no oracle counterpart, so they are emulated from
their specification in include/sfhip.h: index-clamp boundaries == reference
set_bc copies, un_mode 2 == primitive_to_conserved of the source, row range,
wavespeed folding with floor.
"""
import contextlib

import numpy as np

import oracle as _oracle
from sailfish_amd.kernel.system import get_array_module

NG = 2


class _Invocation:
    def __init__(self, lib, name, shape):
        self.lib, self.name, self.shape = lib, name, tuple(shape)

    def bind(self, *args):
        return lambda: self(*args)

    def __call__(self, *args):
        sym = _oracle.API[self.name]
        names = [a.name for a in sym.args][len(self.shape):]
        assert len(names) == len(args), (self.name, len(names), len(args))
        kw = dict(zip(names, args))
        handler = getattr(self.lib, "_" + self.name, None)
        if handler is not None:
            handler(self.shape, kw, list(args))
        else:
            # a reference-named kernel: the oracle has it under the same argument list
            self.lib.orc.call(self.name, self.shape, *[_np(a) if hasattr(a, "numpy") else a for a in args])


class _Kernel:
    def __init__(self, lib, name):
        self.lib, self.name = lib, name

    def __getitem__(self, shape):
        if isinstance(shape, int):
            shape = (shape,)
        return _Invocation(self.lib, self.name, shape)


def _np(t):
    return None if t is None else t.numpy()


class OracleLibrary:
    def __init__(self, orc=None):
        # `orc` may also be one of the compiled-reference libraries (oracle.ref): the reference-named
        # kernels then come from the reference's own C (used by bench.py's cpu_baseline for srhd_2d)
        self.orc = orc or _oracle.oracle()

    def __getattr__(self, name):
        if name in _oracle.API:
            return _Kernel(self, name)
        raise AttributeError(name)

    # -- helpers ----------------------------------------------------------
    @staticmethod
    def _clamped(rd, lo, hi):
        p = rd.copy()
        if lo:
            p[:NG] = p[NG]
        if hi:
            p[-NG:] = p[-NG - 1]
        p[:, :NG] = p[:, NG:NG + 1]
        p[:, -NG:] = p[:, -NG - 1:-NG]
        return p

    def _fold(self, out_t, value, want):
        if want == 2:
            out_t.zero_()
        cur = float(out_t[0])
        out_t[0] = value if (value > cur or value != value) else cur  # max_nan of the kernels: a NaN survives the fold

    def reductions_workspace_len(self, ni, nj):
        return 16

    def _reductions(self, kind, shape, kw):
        """Emulates sfhip_cbd*_2d_reductions from its specification (include/sfhip.h) with the numpy
        restatement of the reference's get_field (tests/ref_solver.py) on the oracle's source-term kernels."""
        from types import SimpleNamespace

        import ref_solver
        from sailfish_amd.physics.circumbinary import Diagnostic, PointMass, SinkModel

        ni, nj = shape
        masses = tuple(
            PointMass(mass=kw[f"mass{n}"], position_x=kw[f"x{n}"], position_y=kw[f"y{n}"], velocity_x=kw[f"vx{n}"],
                      velocity_y=kw[f"vy{n}"], softening_length=kw[f"softening_length{n}"],
                      sink_model=SinkModel(kw[f"sink_model{n}"]), sink_rate=kw[f"sink_rate{n}"],
                      sink_radius=kw[f"sink_radius{n}"]) for n in (1, 2))
        names = {0: 0, 1: 1, 2: 2, 3: 3, 10: "torque", 11: "power", 12: "sigma_m1", 13: "eccentricity_vector",
                 14: "angular_momentum", 15: "mass"}
        spec = _np(kw["spec"]).reshape(-1)[: 8 * kw["num_diagnostics"]].reshape(-1, 8)
        diagnostics = [
            Diagnostic(quantity=names[int(row[0])], which_mass={0: None, 1: 1, 2: 2, 3: "both"}[int(row[1])],
                       gravity=bool(row[2]), accretion=bool(row[3]),
                       radial_cut=(row[5], row[6]) if row[4] else None) for row in spec]
        x, y = _np(kw["xcoord"]), _np(kw["ycoord"])
        dx = (kw["patch_xr"] - kw["patch_xl"]) / ni
        dy = (kw["patch_yr"] - kw["patch_yl"]) / nj
        physics = SimpleNamespace(constant_softening=bool(kw.get("constant_softening", 1)),
                                  gamma_law_index=kw.get("gamma_law_index", 0.0))
        solver = SimpleNamespace(kind=kind, physics=physics, mesh=SimpleNamespace(dx=dx, dy=dy),
                                 _call=lambda name, shp, *a: self.orc.call(f"{kind}_{name}", shp, *a))
        patch = SimpleNamespace(shape=shape, xl=kw["patch_xl"], xr=kw["patch_xr"], yl=kw["patch_yl"], yr=kw["patch_yr"],
                                primitive1=_np(kw["primitive"]),
                                extent=lambda: (kw["patch_xl"], kw["patch_xr"], kw["patch_yl"], kw["patch_yr"]))
        sums = ref_solver.patch_diagnostic_sums(solver, patch, diagnostics, masses=masses, coords=(x, y))
        out = _np(kw["sums"])
        for k, v in enumerate(sums):
            out[2 * k], out[2 * k + 1] = np.real(v), np.imag(v)

    def _sfhip_cbdiso_2d_reductions(self, shape, kw, args):
        self._reductions("cbdiso_2d", shape, kw)

    def _sfhip_cbdgam_2d_reductions(self, shape, kw, args):
        self._reductions("cbdgam_2d", shape, kw)

    def _sfhip_srhd_2d_set_macros(self, shape, kw, args):
        import ctypes

        if self.orc.prefix != "sfo_":  # a compiled reference: its macros are fixed at build time
            if (kw["plm_theta"], kw["mach_ceiling"]) != (1.5, 1e6):
                raise ValueError("the compiled reference srhd_2d library was built with the default macros")
            return
        self.orc.dll.sfo_srhd_2d_set_macros.argtypes = [ctypes.c_double] * 2
        self.orc.dll.sfo_srhd_2d_set_macros(kw["plm_theta"], kw["mach_ceiling"])

    def _sfhip_srhd_2d_stage(self, shape, kw, args):
        """advance_rk, then the conserved_to_primitive call that opens the next stage (include/sfhip.h)."""
        a = {k: (_np(v) if hasattr(v, "numpy") else v) for k, v in kw.items()}
        self.orc.call("srhd_2d_advance_rk", shape, a["face_positions"], a["conserved_rk"], a["primitive_rd"],
                      a["conserved_rd"], a["conserved_wr"], a["polar_extent"], a["a0"], a["adot"], a["time"],
                      a["rk_param"], a["dt"], 0.0, 0.0, 0.0, 0.0, a["num_first_order_zones"])
        a["primitive_wr"][NG:-NG] = a["primitive_rd"][NG:-NG]  # the pressure guess of every zone
        self.orc.call("srhd_2d_conserved_to_primitive", shape, a["face_positions"], a["conserved_wr"],
                      a["conserved_floored"], a["primitive_wr"], a["polar_extent"], a["scale_factor_next"])

    def _sfhip_srhd_2d_failures(self, shape, kw, args):
        kw["out"][0] = float(self.orc.dll.sfo_srhd_2d_failures()) if self.orc.prefix == "sfo_" else 0.0

    def _sfhip_max_nonneg(self, shape, kw, args):
        kw["out"][0] = float(_np(kw["x"]).reshape(-1)[: shape[0]].max())

    @staticmethod
    def _timestep(kw):
        """dt_dx / *amax_device * dt_cfl in IEEE doubles, as the stage kernels form it (include/sfhip.h)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.float64(kw["dt_dx"]) / np.float64(kw["amax_device"][0]) * np.float64(kw["dt_cfl"]))

    def _sfhip_timestep(self, shape, kw, args):
        with np.errstate(divide="ignore"):  # amax == 0 gives inf on the device too
            kw["dt_out"][0] = float(np.float64(kw["dx"]) / np.float64(kw["wavespeed_max"][0]) * np.float64(kw["cfl"]))

    # -- cbdiso -------------------------------------------------------------
    def _sfhip_cbdiso_2d_stage(self, shape, kw, args):
        ni, nj = shape
        rd = self._clamped(_np(kw["primitive_rd"]), kw["clamp_ilo"], kw["clamp_ihi"])
        cons = np.zeros_like(rd)
        if kw["a"] != 0.0 and kw["un_mode"] == 2:
            self.orc.call("cbdiso_2d_primitive_to_conserved", shape, _np(kw["un_source"]).copy(), cons)
        elif kw["a"] != 0.0 and kw["un_mode"] == 1:
            cons = _np(kw["un_source"]).copy()
        out = np.zeros_like(rd)
        ref_args = list(args[:41 - 2])  # the 39 arguments after (ni, nj) shared with cbdiso_2d_advance_rk
        ref_args[4], ref_args[5], ref_args[6] = cons, rd, out
        if kw["dt_device"] is not None:
            ref_args[36] = float(kw["dt_device"][0])
        if kw["amax_device"] is not None:
            ref_args[36] = self._timestep(kw)
        if kw["zero_slot"] is not None:
            kw["zero_slot"][0] = 0.0
        self.orc.call("cbdiso_2d_advance_rk", shape, *ref_args)
        i0, i1 = kw["i_begin"] + NG, kw["i_end"] + NG
        _np(kw["primitive_wr"])[i0:i1, NG:-NG] = out[i0:i1, NG:-NG]
        if kw["want_wavespeed"]:
            ws = np.zeros(rd.shape[:2])
            self.orc.call(
                "cbdiso_2d_wavespeed", shape, *args[0:4], kw["cs2"], kw["mach_squared"], kw["eos_type"],
                kw["wx1"], kw["wy1"], 0.0, 0.0, kw["wmass1"], kw["wsoftening_length1"], 0.0, 0.0, 0,
                kw["wx2"], kw["wy2"], 0.0, 0.0, kw["wmass2"], kw["wsoftening_length2"], 0.0, 0.0, 0, out, ws)
            sel = ws[i0:i1, NG:-NG]
            sel = sel[sel >= kw["wavespeed_floor"]]
            self._fold(kw["wavespeed_max"], float(sel.max()) if sel.size else 0.0, kw["want_wavespeed"])

    def _sfhip_cbdiso_2d_max_wavespeed(self, shape, kw, args):
        prim = _np(kw["primitive"])
        ws = np.zeros(prim.shape[:2])
        self.orc.call(
            "cbdiso_2d_wavespeed", shape, *args[0:7],
            kw["x1"], kw["y1"], 0.0, 0.0, kw["mass1"], kw["softening_length1"], 0.0, 0.0, 0,
            kw["x2"], kw["y2"], 0.0, 0.0, kw["mass2"], kw["softening_length2"], 0.0, 0.0, 0, prim, ws)
        self._fold(kw["wavespeed_max"], float(ws.max()), 2)

    # -- cbdgam -------------------------------------------------------------
    def _sfhip_cbdgam_2d_stage(self, shape, kw, args):
        rd = self._clamped(_np(kw["primitive_rd"]), kw["clamp_ilo"], kw["clamp_ihi"])
        g = kw["gamma_law_index"]
        cons = np.zeros_like(rd)
        if kw["a"] != 0.0 and kw["un_mode"] == 2:
            self.orc.call("cbdgam_2d_primitive_to_conserved", shape, _np(kw["un_source"]).copy(), cons, g)
        elif kw["a"] != 0.0 and kw["un_mode"] == 1:
            cons = _np(kw["un_source"]).copy()
        out = np.zeros_like(rd)
        ref_args = list(args[:44 - 2])
        ref_args[4], ref_args[5], ref_args[6] = cons, rd, out
        if kw["dt_device"] is not None:
            ref_args[35] = float(kw["dt_device"][0])
        if kw["amax_device"] is not None:
            ref_args[35] = self._timestep(kw)
        if kw["zero_slot"] is not None:
            kw["zero_slot"][0] = 0.0
        self.orc.call("cbdgam_2d_advance_rk", shape, *ref_args)
        i0, i1 = kw["i_begin"] + NG, kw["i_end"] + NG
        _np(kw["primitive_wr"])[i0:i1, NG:-NG] = out[i0:i1, NG:-NG]
        if kw["want_wavespeed"]:
            ws = np.zeros(rd.shape[:2])
            self.orc.call("cbdgam_2d_wavespeed", shape, out, ws, g)
            self._fold(kw["wavespeed_max"], float(ws[i0:i1, NG:-NG].max()), kw["want_wavespeed"])

    def _sfhip_cbdgam_2d_max_wavespeed(self, shape, kw, args):
        prim = _np(kw["primitive"])
        ws = np.zeros(prim.shape[:2])
        self.orc.call("cbdgam_2d_wavespeed", shape, prim, ws, kw["gamma_law_index"])
        self._fold(kw["wavespeed_max"], float(ws.max()), 2)

    # -- cbdisodg (one guard zone, weights (ni + 2, nj + 2, 3, 3, 3)) ------------
    @staticmethod
    def _dg_clamped(w, lo, hi, cj):
        w = w.copy()
        if lo:
            w[0] = w[1]
        if hi:
            w[-1] = w[-2]
        if cj:
            w[:, 0] = w[:, 1]
            w[:, -1] = w[:, -2]
        return w

    def _sfhip_cbdisodg_2d_limit(self, shape, kw, args):
        w1 = self._dg_clamped(_np(kw["weights1"]), kw["clamp_lower_i"], kw["clamp_upper_i"], kw["clamp_j"])
        out = np.zeros_like(w1)
        self.orc.call("cbdisodg_2d_slope_limit", shape, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, w1, out)
        i0, i1 = kw["i_begin"] + 1, kw["i_end"] + 1
        _np(kw["weights2"])[i0:i1, 1:-1] = out[i0:i1, 1:-1]

    def _sfhip_cbdisodg_2d_stage(self, shape, kw, args):
        w1 = self._dg_clamped(_np(kw["weights1"]), kw["clamp_lower_i"], kw["clamp_upper_i"], kw["clamp_j"])
        out = _np(kw["weights2"]).copy()  # the three slots outside the scheme keep what the buffer held
        ref_args = list(args[:40 - 2])  # the 38 arguments after (ni, nj) shared with cbdisodg_2d_advance_rk
        ref_args[4], ref_args[5], ref_args[6] = _np(kw["weights0"]).copy(), w1, out
        if kw["dt_device"] is not None:
            ref_args[36] = float(kw["dt_device"][0])
        self.orc.call("cbdisodg_2d_advance_rk", shape, *ref_args)
        i0, i1 = kw["i_begin"] + 1, kw["i_end"] + 1
        _np(kw["weights2"])[i0:i1, 1:-1] = out[i0:i1, 1:-1]
        if kw["want_wavespeed"]:
            ws = np.zeros(w1.shape[:2])
            self.orc.call(
                "cbdisodg_2d_wavespeed", shape, *args[0:4], kw["soundspeed2"], kw["mach_squared"], kw["eos_type"],
                kw["wx1"], kw["wy1"], 0.0, 0.0, kw["wmass1"], kw["wsoftening_length1"], 0.0, 0.0, 0,
                kw["wx2"], kw["wy2"], 0.0, 0.0, kw["wmass2"], kw["wsoftening_length2"], 0.0, 0.0, 0,
                kw["velocity_ceiling"], out, ws)
            self._fold(kw["wavespeed_max"], float(ws[i0:i1, 1:-1].max()), kw["want_wavespeed"])

    def _sfhip_cbdisodg_2d_max_wavespeed(self, shape, kw, args):
        w = _np(kw["weights"])
        ws = np.zeros(w.shape[:2])
        self.orc.call(
            "cbdisodg_2d_wavespeed", shape, *args[0:7],
            kw["x1"], kw["y1"], 0.0, 0.0, kw["mass1"], kw["softening_length1"], 0.0, 0.0, 0,
            kw["x2"], kw["y2"], 0.0, 0.0, kw["mass2"], kw["softening_length2"], 0.0, 0.0, 0,
            kw["velocity_ceiling"], w, ws)
        self._fold(kw["wavespeed_max"], float(ws.max()), 2)


class OracleBackend:
    mode = "host"

    def __init__(self, orc=None):
        self.lib = OracleLibrary(orc)
        self.xp = get_array_module("host")

    def context(self, device_id):
        return contextlib.nullcontext()

    def num_devices(self):
        return 1
