"""
Pins the oracle: our C restatement (oracle/sf_oracle.c) must reproduce the
reference's own C kernels (oracle/_ref, compiled in place from /root/reference)
BIT FOR BIT on every branch combination.  Skipped only where neither
/root/reference nor a prebuilt oracle/_ref is present.
"""
import numpy as np
import pytest

import cases
import oracle

ORC = oracle.oracle()
REF_ISO = oracle.ref("cbdiso_2d")
REF_GAM = oracle.ref("cbdgam_2d")
need_iso = pytest.mark.skipif(REF_ISO is None, reason="compiled reference (oracle/_ref) not available")
need_gam = pytest.mark.skipif(REF_GAM is None, reason="compiled reference (oracle/_ref) not available")


@need_iso
@pytest.mark.parametrize("kw", cases.iso_branch_matrix(), ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_iso_advance_bitwise(kw):
    c = cases.iso_case(seed=7, **kw)
    out_o = np.full_like(c["prim"], -7.0)
    out_r = np.full_like(c["prim"], -7.0)
    ORC.call("cbdiso_2d_advance_rk", c["shape"], *cases.iso_advance_args(c, c["cons"], c["prim"], out_o))
    REF_ISO.call("cbdiso_2d_advance_rk", c["shape"], *cases.iso_advance_args(c, c["cons"], c["prim"], out_r))
    assert np.array_equal(out_o, out_r)
    assert np.all(out_o[:2] == -7.0) and np.all(out_o[:, :2] == -7.0)  # guards untouched
    # the recovery's limits (cbdiso_2d.c:274-278) are really taken by the cases that ask for them
    floored, capped = cases.iso_limits_engaged(c, out_r)
    zones = c["shape"][0] * c["shape"][1]
    assert floored >= zones // 50 if "dfloor" in kw else floored == 0, floored
    assert capped >= zones // 50 if "vmax" in kw else capped == 0, capped


@need_iso
@pytest.mark.parametrize("eos", [1, 2])
def test_iso_wavespeed_p2c_rate_bitwise(eos):
    c = cases.iso_case(seed=3, eos=eos, ni=40, nj=23)
    for name, args_of in (
        ("cbdiso_2d_wavespeed", lambda out: cases.iso_wavespeed_args(c, c["prim"], out)),
        ("cbdiso_2d_primitive_to_conserved", lambda out: (c["prim"], out)),
        ("cbdiso_2d_point_mass_source_term", lambda out: (*c["extent"], *c["m1"], c["prim"], out)),
    ):
        shape = c["prim"].shape[:2] if name.endswith("wavespeed") else c["prim"].shape
        a, b = np.zeros(shape), np.zeros(shape)
        ORC.call(name, c["shape"], *args_of(a))
        REF_ISO.call(name, c["shape"], *args_of(b))
        assert np.array_equal(a, b), name


@need_gam
@pytest.mark.parametrize("kw", cases.gam_branch_matrix(), ids=lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()))
def test_gam_advance_bitwise(kw):
    c = cases.gam_case(seed=11, **kw)
    out_o = np.full_like(c["prim"], -7.0)
    out_r = np.full_like(c["prim"], -7.0)
    ORC.call("cbdgam_2d_advance_rk", c["shape"], *cases.gam_advance_args(c, c["cons"], c["prim"], out_o))
    REF_GAM.call("cbdgam_2d_advance_rk", c["shape"], *cases.gam_advance_args(c, c["cons"], c["prim"], out_r))
    assert np.array_equal(out_o, out_r, equal_nan=True)
    # the limits of the recovery and of the cooling term (cbdgam_2d.c:314, 328-339) are really taken
    reset, pfloored, capped = cases.gam_limits_engaged(c, out_r)
    zones = c["shape"][0] * c["shape"][1]
    assert reset >= zones // 50 if "dfloor" in kw else reset == 0, reset
    assert pfloored >= zones // 50 if "pfloor" in kw else pfloored == 0, pfloored
    assert capped >= zones // 50 if "vmax" in kw else capped == 0, capped
    if "mach_ceiling" in kw:
        free = cases.gam_case(seed=11, **dict(kw, mach_ceiling=1e5))
        out_f = np.full_like(c["prim"], -7.0)
        REF_GAM.call("cbdgam_2d_advance_rk", c["shape"], *cases.gam_advance_args(free, c["cons"], c["prim"], out_f))
        raised = (out_r[2:-2, 2:-2, 3] > out_f[2:-2, 2:-2, 3]).sum()
        assert raised > zones // 10, raised  # the Mach ceiling lifted the internal energy of these zones


@need_gam
def test_gam_wavespeed_p2c_rate_bitwise():
    c = cases.gam_case(seed=5, ni=40, nj=23)
    g = c["gamma"]
    for name, shape, args_of in (
        ("cbdgam_2d_wavespeed", c["prim"].shape[:2], lambda out: (c["prim"], out, g)),
        ("cbdgam_2d_primitive_to_conserved", c["prim"].shape, lambda out: (c["prim"], out, g)),
        ("cbdgam_2d_point_mass_source_term", c["prim"].shape,
         lambda out: (*c["extent"], *c["m1"], *c["m2"], 2, c["prim"], out, 0, g)),
    ):
        a, b = np.zeros(shape), np.zeros(shape)
        ORC.call(name, c["shape"], *args_of(a))
        REF_GAM.call(name, c["shape"], *args_of(b))
        assert np.array_equal(a, b), name


# ---------------------------------------------------------------------------
# srhd_2d (oracle/sf_oracle_srhd.c); two compiled references: default macros, and
# PLM_THETA=2.0 / MACH_CEILING=100 to pin the oracle's run-time settings
# ---------------------------------------------------------------------------
REF_SRHD = {(1.5, 1e6): oracle.ref("srhd_2d"), (2.0, 100.0): oracle.ref("srhd_2d", "cpu_theta2")}
need_srhd = pytest.mark.skipif(REF_SRHD[(1.5, 1e6)] is None, reason="compiled reference (oracle/_ref) not available")
SRHD_IDS = lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()) or "default"  # noqa: E731


@need_srhd
@pytest.mark.parametrize("macros", list(REF_SRHD))
@pytest.mark.parametrize("kw", cases.srhd_branch_matrix(), ids=SRHD_IDS)
def test_srhd_advance_bitwise(kw, macros):
    ref = REF_SRHD[macros]
    ORC.dll.sfo_srhd_2d_set_macros.argtypes = [oracle.ctypes.c_double] * 2
    ORC.dll.sfo_srhd_2d_set_macros(*macros)
    try:
        c = cases.srhd_case(seed=13, **kw)
        cons_rd = cases.srhd_conserved(ref, c)
        cons_rk = cases.srhd_conserved(ref, c, prim=c["prim"] * 1.01)
        out_o = np.full_like(c["prim"], -7.0)
        out_r = np.full_like(c["prim"], -7.0)
        ORC.call("srhd_2d_advance_rk", c["shape"], *cases.srhd_advance_args(c, cons_rk, c["prim"], cons_rd, out_o))
        ref.call("srhd_2d_advance_rk", c["shape"], *cases.srhd_advance_args(c, cons_rk, c["prim"], cons_rd, out_r))
        assert np.array_equal(out_o, out_r)
        assert np.all(out_o[:2] == -7.0) and np.all(out_o[-2:] == -7.0)  # guards untouched
        if c["jet"][0] > 0.0 and c["time"] > 6.0:
            assert np.all(out_o[2] == -7.0)  # jet off: innermost row is not written at all
    finally:
        ORC.dll.sfo_srhd_2d_set_macros(1.5, 1e6)


@need_srhd
@pytest.mark.parametrize("macros", list(REF_SRHD))
@pytest.mark.parametrize("cold", [False, True])
@pytest.mark.parametrize("adot", [0.0, 0.15])
def test_srhd_p2c_c2p_wavespeed_bitwise(macros, cold, adot):
    ref = REF_SRHD[macros]
    ORC.dll.sfo_srhd_2d_set_macros.argtypes = [oracle.ctypes.c_double] * 2
    ORC.dll.sfo_srhd_2d_set_macros(*macros)
    try:
        c = cases.srhd_case(seed=21, ni=30, nj=17, adot=adot, time=1.4, cold=cold)
        cons_o, cons_r = cases.srhd_conserved(ORC, c), cases.srhd_conserved(ref, c)
        assert np.array_equal(cons_o, cons_r)
        rng = np.random.default_rng(5)
        guess = c["prim"].copy()
        guess[..., 3] *= rng.uniform(0.7, 1.4, guess.shape[:2])
        results = []
        for lib in (ORC, ref):
            p = guess.copy()
            u2 = np.full_like(cons_r, -3.0)
            lib.call("srhd_2d_conserved_to_primitive", c["shape"], c["faces"], cons_r, u2, p, c["polar_extent"],
                     c["scale_factor"])
            w = np.full(c["shape"], -1.0)
            lib.call("srhd_2d_max_wavespeeds", c["shape"], c["faces"], p, w, c["adot"])
            results.append((p, u2, w))
        for a, b in zip(*results):
            assert np.array_equal(a, b)
        p, u2, w = results[0]
        floored = np.any(u2[2:-2] != cons_r[2:-2], axis=-1)
        assert floored.any() == (cold and macros[1] == 100.0)  # the Mach-ceiling floor engages only there
        if not cold:
            assert np.allclose(p[2:-2], c["prim"][2:-2], rtol=1e-9)
        assert ORC.dll.sfo_srhd_2d_failures() == 0
    finally:
        ORC.dll.sfo_srhd_2d_set_macros(1.5, 1e6)


# ---------------------------------------------------------------------------
# cbdisodg_2d (oracle/sf_oracle_dg.c)
# ---------------------------------------------------------------------------
REF_DG = oracle.ref("cbdisodg_2d")
need_dg = pytest.mark.skipif(REF_DG is None, reason="compiled reference (oracle/_ref) not available")


@need_dg
@pytest.mark.parametrize("kw", cases.dg_branch_matrix(), ids=SRHD_IDS)
def test_dg_advance_bitwise(kw):
    c = cases.dg_case(seed=17, **kw)
    out_o = np.full_like(c["w1"], -7.0)
    out_r = np.full_like(c["w1"], -7.0)
    ORC.call("cbdisodg_2d_advance_rk", c["shape"], *cases.dg_advance_args(c, c["w0"], c["w1"], out_o))
    REF_DG.call("cbdisodg_2d_advance_rk", c["shape"], *cases.dg_advance_args(c, c["w0"], c["w1"], out_r))
    assert np.array_equal(out_o, out_r)
    assert np.all(out_o[0] == -7.0) and np.all(out_o[:, -1] == -7.0)  # guards untouched
    for m, n in ((1, 2), (2, 1), (2, 2)):
        assert np.all(out_o[..., m, n] == -7.0)  # slots outside the scheme are not written
    assert not np.any(out_o[1:-1, 1:-1, :, 0, 0] == -7.0)


@need_dg
@pytest.mark.parametrize("kw", cases.dg_limit_matrix(), ids=SRHD_IDS)
def test_dg_slope_limit_bitwise(kw):
    c = cases.dg_case(seed=23, **kw)
    out_o = np.full_like(c["w1"], -7.0)
    out_r = np.full_like(c["w1"], -7.0)
    ORC.call("cbdisodg_2d_slope_limit", c["shape"], *cases.dg_limit_args(c, c["w1"], out_o))
    REF_DG.call("cbdisodg_2d_slope_limit", c["shape"], *cases.dg_limit_args(c, c["w1"], out_r))
    assert np.array_equal(out_o, out_r)
    inner_in, inner_out = c["w1"][1:-1, 1:-1], out_o[1:-1, 1:-1]
    assert np.array_equal(inner_out[..., 0, 0], inner_in[..., 0, 0])  # zone averages are never touched
    limited = np.any(inner_out != inner_in, axis=(-1, -2))
    if kw.get("moments", 1.0) <= 0.005:
        assert limited.any() and not limited.all()  # both sides of the troubled-cell test
    if kw.get("big_slopes"):
        slopes = inner_out[limited][..., 1, 0]
        assert np.all(slopes == np.trunc(slopes)) and np.any(slopes != 0.0)  # the reference's `int` return


@need_dg
@pytest.mark.parametrize("eos", [1, 2])
def test_dg_wavespeed_bitwise(eos):
    c = cases.dg_case(seed=3, eos=eos, ni=40, nj=23)
    a, b = np.zeros(c["w1"].shape[:2]), np.zeros(c["w1"].shape[:2])
    ORC.call("cbdisodg_2d_wavespeed", c["shape"], *cases.dg_wavespeed_args(c, c["w1"], a))
    REF_DG.call("cbdisodg_2d_wavespeed", c["shape"], *cases.dg_wavespeed_args(c, c["w1"], b))
    assert np.array_equal(a, b) and a[1:-1, 1:-1].min() > 0.0
