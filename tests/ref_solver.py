"""
Test infrastructure: the reference's solver loop, in the reference's own order
of operations, on numpy arrays with CPU kernels from the oracle
(oracle/libsf_oracle.so) or from the compiled reference (oracle/_ref).

Per step (sailfish/solvers/cbdiso_2d.py:536-551, cbdgam_2d.py:473-481):
    new_iteration: time0 = time; conserved0 = p2c(primitive1)
    for a in rk parameters: set_bc(primitive1); advance_rk(a, dt); swap
and ``maximum_wavespeed`` = wavespeed kernel + array max over all patches.

Used as the expected result for solver-level GPU parity tests, by smoke(), and
as the timed CPU baseline in bench.py.  Never imported by sailfish_amd.
"""
import numpy as np

import oracle as _oracle
from sailfish_amd.physics.circumbinary import Physics
from sailfish_amd.subdivide import subdivide

NG = 2


class OraclePatch:
    def __init__(self, solver, prim, index_range):
        self.s = solver
        mesh = solver.mesh
        i0, i1 = index_range
        self.shape = (i1 - i0, mesh.shape[1])
        self.xl, self.yl = mesh.vertex_coordinates(i0, 0)
        self.xr, self.yr = mesh.vertex_coordinates(i1, mesh.shape[1])
        self.time = self.time0 = solver.time0
        self.primitive1 = prim.copy()
        self.primitive2 = prim.copy()
        self.conserved0 = np.zeros_like(prim)
        self.wavespeeds = np.zeros(prim.shape[:2])

    def extent(self):
        return (self.xl, self.xr, self.yl, self.yr)


class OracleSolver:
    def __init__(self, setup, mesh, time=0.0, num_patches=1, lib=None, options=None, solution=None):
        self.setup = setup
        self.mesh = mesh
        self.kind = setup.solver  # cbdiso_2d | cbdgam_2d
        self.lib = lib or _oracle.oracle()
        self.physics = Physics(**{k: v for k, v in setup.physics.items() if k != "diagnostics"})
        self.time0 = time
        ph = self.physics
        nq = 3 if self.kind == "cbdiso_2d" else 4
        self.nq = nq
        opts = dict(options or {})
        if self.kind == "cbdiso_2d":
            self.opt = dict(velocity_ceiling=1e12, density_floor=1e-12, rk_order=2)
        else:
            self.opt = dict(pressure_floor=1e-12, density_floor=1e-10, velocity_ceiling=1e16, mach_ceiling=1e5,
                            rk_order=2)
        self.opt.update(opts)
        if ph.buffer_is_enabled:
            sample = [0.0] * nq
            self.buffer_outer_radius = mesh.x1
            setup.primitive(time, [mesh.x1 - ph.buffer_onset_width, 0.0], sample)
            self.buffer_surface_density = sample[0]
            self.buffer_surface_pressure = sample[3] if nq == 4 else 0.0
        else:
            self.buffer_outer_radius = self.buffer_surface_density = self.buffer_surface_pressure = 0.0
        ni, nj = mesh.shape
        self.patches = []
        for a, b in subdivide(ni, num_patches):
            prim = np.zeros([b - a + 2 * NG, nj + 2 * NG, nq])
            prim[NG:-NG, NG:-NG] = solution[a:b] if solution is not None else setup.primitive_array(time, mesh, a, b, nq)
            self.patches.append(OraclePatch(self, prim, (a, b)))

    # ------------------------------------------------------------------
    @property
    def time(self):
        return self.patches[0].time

    @property
    def solution(self):
        return np.concatenate([p.primitive1[NG:-NG, NG:-NG] for p in self.patches], axis=0)

    def _call(self, name, shape, *args):
        self.lib.call(f"{self.kind}_{name}", shape, *args)

    def maximum_wavespeed(self):
        ph = self.physics
        result = []
        for p in self.patches:
            if self.kind == "cbdiso_2d":
                m1, m2 = ph.point_masses(p.time)
                self._call("wavespeed", p.shape, *p.extent(), ph.sound_speed**2, ph.mach_number**2,
                           ph.eos_type.value, *m1.kernel_args(), *m2.kernel_args(), p.primitive1, p.wavespeeds)
            else:
                self._call("wavespeed", p.shape, p.primitive1, p.wavespeeds, ph.gamma_law_index)
            result.append(float(p.wavespeeds.max()))
        return max(result)

    def set_bc(self):
        n = len(self.patches)
        for k, pk in enumerate(self.patches):
            pl = self.patches[(k + n - 1) % n].primitive1
            pr = self.patches[(k + 1) % n].primitive1
            pc = pk.primitive1
            pc[:NG] = pl[-2 * NG:-NG]
            pc[-NG:] = pr[NG:2 * NG]
            if k == 0:
                pc[:NG] = pc[NG]
            if k == n - 1:
                pc[-NG:] = pc[-NG - 1]
            pc[:, :NG] = pc[:, NG:NG + 1]
            pc[:, -NG:] = pc[:, -NG - 1:-NG]

    def advance_rk(self, a, dt):
        ph, o = self.physics, self.opt
        self.set_bc()
        for p in self.patches:
            m1, m2 = ph.point_masses(p.time)
            if self.kind == "cbdiso_2d":
                self._call(
                    "advance_rk", p.shape, *p.extent(), p.conserved0, p.primitive1, p.primitive2,
                    self.buffer_surface_density, m1.mass + m2.mass, ph.buffer_driving_rate,
                    self.buffer_outer_radius, ph.buffer_onset_width, int(ph.buffer_is_enabled),
                    *m1.kernel_args(), *m2.kernel_args(),
                    ph.sound_speed**2, ph.mach_number**2, ph.eos_type.value, ph.viscosity_coefficient,
                    a, dt, o["velocity_ceiling"], o["density_floor"])
            else:
                self._call(
                    "advance_rk", p.shape, *p.extent(), p.conserved0, p.primitive1, p.primitive2,
                    ph.gamma_law_index, self.buffer_surface_density, self.buffer_surface_pressure,
                    m1.mass + m2.mass, ph.buffer_driving_rate, self.buffer_outer_radius, ph.buffer_onset_width,
                    int(ph.buffer_is_enabled), *m1.kernel_args(), *m2.kernel_args(),
                    ph.alpha, a, dt, o["velocity_ceiling"], ph.cooling_coefficient, o["mach_ceiling"],
                    o["density_floor"], o["pressure_floor"], int(ph.constant_softening))
            p.time = p.time0 * a + (p.time + dt) * (1.0 - a)
            p.primitive1, p.primitive2 = p.primitive2, p.primitive1

    def advance(self, dt):
        for p in self.patches:
            p.time0 = p.time
            if self.kind == "cbdiso_2d":
                self._call("primitive_to_conserved", p.shape, p.primitive1, p.conserved0)
            else:
                self._call("primitive_to_conserved", p.shape, p.primitive1, p.conserved0, self.physics.gamma_law_index)
        for a in {1: [0.0], 2: [0.0, 0.5], 3: [0.0, 0.75, 1.0 / 3.0]}[self.opt["rk_order"]]:
            self.advance_rk(a, dt)


    # ------------------------------------------------------------------ diagnostics
    def reductions(self, diagnostics):
        """
        Solver.reductions() of the reference, restated on numpy (cbdiso_2d.py:370-503,
        cbdgam_2d.py:341-443): same temporaries, same order of operations, `f.sum()` per patch,
        Python `sum` over patches, then `* da`.  `diagnostics` is a list of Diagnostic tuples.
        """
        da = self.mesh.dx * self.mesh.dy
        out = []
        per_patch = [patch_diagnostic_sums(self, p, diagnostics) for p in self.patches]
        for k, d in enumerate(diagnostics):
            if d.quantity == "time":
                out.append(self.time / self.setup.reference_time_scale)
            else:
                out.append(sum(sums[k] for sums in per_patch) * da)
        return out


def patch_source_term(solver, p, which_mass, gravity=False, accretion=False, masses=None):
    """Patch.point_mass_source_term (cbdiso_2d.py:105-137, cbdgam_2d.py:104-143) on the CPU kernels."""
    m1, m2 = masses or solver.physics.point_masses(p.time)
    rate = np.zeros_like(p.primitive1)

    def scaled(m):
        return (m.position_x, m.position_y, m.velocity_x, m.velocity_y, m.mass * gravity, m.softening_length,
                m.sink_rate * accretion, m.sink_radius, m.sink_model.value)

    if solver.kind == "cbdiso_2d":
        solver._call("point_mass_source_term", p.shape, *p.extent(), *scaled((m1, m2)[which_mass - 1]),
                     p.primitive1, rate)
    else:
        solver._call("point_mass_source_term", p.shape, *p.extent(), *scaled(m1), *scaled(m2), which_mass,
                     p.primitive1, rate, int(solver.physics.constant_softening), solver.physics.gamma_law_index)
    return rate[NG:-NG, NG:-NG]


def patch_diagnostic_sums(solver, p, diagnostics, masses=None, coords=None, absolute=False):
    """One patch's `f.sum()` for every diagnostic (None for "time"); get_field of the reference, literally."""
    mesh = solver.mesh
    ni, nj = p.shape
    # Patch.cell_center_coordinate_arrays (cbdiso_2d.py:82-89, 96-103)
    if coords is None:
        x = np.linspace(p.xl + 0.5 * mesh.dx, p.xr - 0.5 * mesh.dx, ni)[:, None]
        y = np.linspace(p.yl + 0.5 * mesh.dy, p.yr - 0.5 * mesh.dy, nj)[None, :]
    else:
        x, y = coords[0].reshape(-1, 1), coords[1].reshape(1, -1)
    m1, m2 = masses or solver.physics.point_masses(p.time)
    acc = [patch_source_term(solver, p, w, accretion=True, masses=(m1, m2)) for w in (1, 2)]
    grv = [patch_source_term(solver, p, w, gravity=True, masses=(m1, m2)) for w in (1, 2)]
    prim = p.primitive1[NG:-NG, NG:-NG]
    r = (x**2 + y**2) ** 0.5

    def get_field(quantity, cut, mass, gravity, accretion):
        def apply_radial_cut(f):
            if cut is not None:
                r0, r1 = cut
                return f * (r0 < r) * (r < r1)
            return f

        if quantity == "mdot":
            return get_field(0, cut, mass, gravity, accretion)
        if quantity == "fx":
            return get_field(1, cut, mass, gravity, accretion)
        if quantity == "fy":
            return get_field(2, cut, mass, gravity, accretion)
        if quantity == "torque":
            fx = get_field(1, cut, mass, gravity, accretion)
            fy = get_field(2, cut, mass, gravity, accretion)
            return x * fy - y * fx
        if quantity == "sigma_m1":
            sigma = apply_radial_cut(prim[..., 0])
            return sigma * (x / r + 1.0j * (y / r))
        if quantity == "eccentricity_vector":
            sigma = apply_radial_cut(prim[..., 0])
            vx = apply_radial_cut(prim[..., 1])
            vy = apply_radial_cut(prim[..., 2])
            GM = 1.0
            v_dot_v = vx * vx + vy * vy
            v_dot_r = vx * x + vy * y
            ex = (v_dot_v * x - v_dot_r * vx) / GM - x / r
            ey = (v_dot_v * y - v_dot_r * vy) / GM - y / r
            return sigma * (ex + 1.0j * ey)
        if quantity == "angular_momentum":
            sigma = apply_radial_cut(prim[..., 0])
            vx = apply_radial_cut(prim[..., 1])
            vy = apply_radial_cut(prim[..., 2])
            return sigma * (x * vy - y * vx)
        if quantity == "mass":
            return apply_radial_cut(prim[..., 0])
        if quantity == "power":
            fx = get_field(1, cut, mass, gravity, accretion)
            fy = get_field(2, cut, mass, gravity, accretion)
            if mass not in (1, 2):
                raise ValueError("Mass option for 'power' must be 1 or 2.")
            m = (m1, m2)[mass - 1]
            return m.velocity_x * fx + m.velocity_y * fy
        udots = acc if accretion else grv
        if mass == "both":
            f = udots[0][..., quantity] + udots[1][..., quantity]
        else:
            f = udots[mass - 1][..., quantity]
        return apply_radial_cut(f)

    sums = []
    for d in diagnostics:
        if d.quantity == "time":
            sums.append(None)
        else:
            f = get_field(d.quantity, d.radial_cut, d.which_mass, d.gravity, d.accretion)
            sums.append(np.abs(f).sum() if absolute else f.sum())
    return sums
