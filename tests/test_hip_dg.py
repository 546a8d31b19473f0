"""
GPU parity of the cbdisodg_2d entry points (sailfish_amd/csrc/sf_dg.hpp) against the CPU oracle
(oracle/sf_oracle_dg.c, itself bit-identical to the compiled reference).

Bars: the device kernels perform the reference's IEEE operations in the reference's order, so
the limiter, the wavespeed, and the update with the point masses switched off must be BIT-EXACT
(`==` on every double; the kernels may differ from the reference in the sign of an exact zero,
see sf_dg.hpp).  With point masses the update evaluates pow(x, -1.5) and exp() on the device:
held to 1e-12 of the zone's largest weight.
"""
import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu
TOL = 1e-12
IDS = lambda kw: ",".join(f"{k}={v}" for k, v in kw.items()) or "default"  # noqa: E731


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def orc():
    return oracle.oracle()


def zone_error(got, expect):
    """max over zones of |got - expect|_inf / |expect|_inf, both over the zone's 27 weights"""
    num = np.abs(got - expect).reshape(*got.shape[:2], -1).max(axis=-1)
    den = np.abs(expect).reshape(*got.shape[:2], -1).max(axis=-1)
    return float((num / den).max())


@pytest.mark.parametrize("kw", cases.dg_branch_matrix(), ids=IDS)
def test_dg_advance(kw, orc, hip_lib):
    c = cases.dg_case(seed=17, **kw)
    expect = np.full_like(c["w1"], -7.0)
    orc.call("cbdisodg_2d_advance_rk", c["shape"], *cases.dg_advance_args(c, c["w0"], c["w1"], expect))
    d_out = dev(np.full_like(c["w1"], -7.0))
    hip_lib.cbdisodg_2d_advance_rk[c["shape"]](*cases.dg_advance_args(c, dev(c["w0"]), dev(c["w1"]), d_out))
    got = host(d_out)
    assert np.array_equal(got == -7.0, expect == -7.0)  # guards and the three unused slots stay untouched
    if kw.get("masses", True):
        assert zone_error(got, expect) <= TOL
    else:
        assert np.array_equal(got, expect)


@pytest.mark.parametrize("kw", cases.dg_limit_matrix(), ids=IDS)
def test_dg_slope_limit_bit_exact(kw, orc, hip_lib):
    c = cases.dg_case(seed=23, **kw)
    expect = np.full_like(c["w1"], -7.0)
    orc.call("cbdisodg_2d_slope_limit", c["shape"], *cases.dg_limit_args(c, c["w1"], expect))
    d_out = dev(np.full_like(c["w1"], -7.0))
    hip_lib.cbdisodg_2d_slope_limit[c["shape"]](*cases.dg_limit_args(c, dev(c["w1"]), d_out))
    assert np.array_equal(host(d_out), expect)


@pytest.mark.parametrize("eos", [1, 2])
def test_dg_wavespeed_bit_exact(eos, orc, hip_lib):
    c = cases.dg_case(seed=3, eos=eos, ni=40, nj=150)
    expect = np.zeros(c["w1"].shape[:2])
    orc.call("cbdisodg_2d_wavespeed", c["shape"], *cases.dg_wavespeed_args(c, c["w1"], expect))
    d_w, d_out, d_max = dev(c["w1"]), dev(np.zeros_like(expect)), dev(np.full(1, -1.0))
    hip_lib.cbdisodg_2d_wavespeed[c["shape"]](*cases.dg_wavespeed_args(c, d_w, d_out))
    assert np.array_equal(host(d_out), expect)
    m1, m2 = c["m1"], c["m2"]
    hip_lib.sfhip_cbdisodg_2d_max_wavespeed[c["shape"]](
        *c["extent"], *c["eos"], m1[0], m1[1], m1[4], m1[5], m2[0], m2[1], m2[4], m2[5], c["vmax"], d_w, d_max)
    assert host(d_max)[0] == expect.max()


def fill_guards(w):
    """Solver.set_bc for a single patch (cbdisodg_2d.py:522-544)."""
    w = w.copy()
    w[0], w[-1] = w[1], w[-2]
    w[:, 0], w[:, -1] = w[:, 1], w[:, -2]
    return w


@pytest.mark.parametrize("kw", [dict(masses=False, buffer_on=False, nu=0.01, rk_param=0.5, eos=1),
                                dict(ni=36, nj=130, masses=False, rk_param=0.5, moments=0.003)], ids=IDS)
def test_dg_clamped_rows_and_epilogue(kw, orc, hip_lib):
    """
    sfhip_cbdisodg_2d_limit / _stage with index clamps and row ranges reproduce set_bc + the
    reference kernels bit for bit without ever reading a guard zone (poisoned here), the device-side
    dt is the one used, and the wavespeed epilogue equals cbdisodg_2d_wavespeed + max of the output.
    """
    c = cases.dg_case(seed=31, **kw)
    ni, nj = c["shape"]
    w1 = fill_guards(c["w1"])
    limited = np.zeros_like(w1)
    orc.call("cbdisodg_2d_slope_limit", c["shape"], *cases.dg_limit_args(c, w1, limited))
    limited = fill_guards(limited)
    expect = np.zeros_like(w1)
    orc.call("cbdisodg_2d_advance_rk", c["shape"], *cases.dg_advance_args(c, c["w0"], limited, expect))
    speeds = np.zeros(w1.shape[:2])
    orc.call("cbdisodg_2d_wavespeed", c["shape"], *cases.dg_wavespeed_args(c, expect, speeds))

    poisoned = c["w1"].copy()
    poisoned[0] = poisoned[-1] = np.nan
    poisoned[:, 0] = poisoned[:, -1] = np.nan
    d_w1, d_lim, d_out = dev(poisoned), dev(np.full_like(w1, np.nan)), dev(np.zeros_like(w1))
    split = ni // 2
    for lo, hi in ((0, split), (split, ni)):
        hip_lib.sfhip_cbdisodg_2d_limit[c["shape"]](d_w1, d_lim, 1, 1, 1, lo, hi)
    assert np.array_equal(host(d_lim)[1:-1, 1:-1], limited[1:-1, 1:-1])
    d_max, d_dt = dev(np.full(1, 123.0)), dev(np.full(1, c["dt"]))
    m1, m2 = c["m1"], c["m2"]
    args = cases.dg_advance_args(c, dev(c["w0"]), d_lim, d_out)
    args = args[:-2] + (float("nan"), args[-1])  # the host-side dt must not be looked at
    for lo, hi, want in ((0, split, 2), (split, ni, 1)):
        hip_lib.sfhip_cbdisodg_2d_stage[c["shape"]](
            *args, 1, 1, 1, lo, hi, want, m1[0], m1[1], m1[4], m1[5], m2[0], m2[1], m2[4], m2[5], d_max, d_dt)
    assert np.array_equal(host(d_out)[1:-1, 1:-1], expect[1:-1, 1:-1])
    assert host(d_max)[0] == speeds.max()


@pytest.mark.parametrize("kw", [dict(ni=21, nj=130, extent=(-1.7, 1.3, -2.1, 2.9)),
                                dict(ni=9, nj=63, extent=(-1.0, 2.0, -0.7, 0.9), nu=0.01, rk_param=0.5),
                                dict(ni=6, nj=64, eos=1, extent=(-1.1, 1.1, -1.3, 1.3)),
                                dict(ni=33, nj=200, extent=(-2.0, 2.0, -2.0, 2.0), sink_model=1),
                                # a wave stages 65 zones of a row at once: a row that ends one zone short of that, and
                                # one whose last wave's 65 zones end exactly at the row's end
                                dict(ni=7, nj=62, extent=(-1.3, 1.1, -0.9, 1.9), nu=0.01),
                                dict(ni=8, nj=126, extent=(-1.9, 1.4, -1.2, 2.3), rk_param=0.5)], ids=IDS)
@pytest.mark.parametrize("clamped", [False, True], ids=["guards", "clamped"])
@pytest.mark.parametrize("rows_per_wave", [1, 3, 8])
def test_dg_faces_solved_once_equal_faces_solved_twice(kw, clamped, rows_per_wave, hip_lib, monkeypatch):
    """
    dg_advance_kernel solves a face once for the zones on both sides of it, wherever both belong to the same wave's
    march (sf_dg.hpp; every `rows_per_wave` rows a face is still solved twice); the only thing the two sides can
    disagree on is the position at which the sound speed is taken (locally isothermal gas, mesh spacing that is not
    a power of two: the cases here).  With the point masses on, the oracle pins the update to 1e-12 only
    (device pow / exp), so this property is pinned against the kernel of rounds 1-2, in which every zone solves its
    own four faces as the reference's zones do (SFHIP_DG_ZONE_KERNEL=1): same device functions, so `==`,
    on whole patches and on row ranges, with guard zones and with index clamps.
    """
    c = cases.dg_case(seed=41, **kw)
    ni, nj = c["shape"]
    w1 = c["w1"] if not clamped else fill_guards(c["w1"])
    m1, m2 = c["m1"], c["m2"]
    flags = (1, 1, 1) if clamped else (0, 0, 0)
    outs = []
    for zone_kernel in ("1", "0"):
        monkeypatch.setenv("SFHIP_DG_ZONE_KERNEL", zone_kernel)
        monkeypatch.setenv("SFHIP_DG_ROWS_PER_WAVE", str(rows_per_wave))
        src = w1.copy()
        if clamped:  # the clamps must keep both kernels away from the guard zones
            src[0] = src[-1] = np.nan
            src[:, 0] = src[:, -1] = np.nan
        d_out, d_max, d_dt = dev(np.full_like(w1, -7.0)), dev(np.zeros(1)), dev(np.full(1, c["dt"]))
        args = cases.dg_advance_args(c, dev(c["w0"]), dev(src), d_out)
        for lo, hi, want in ((0, 5, 2), (5, ni - 1, 1), (ni - 1, ni, 1)):
            hip_lib.sfhip_cbdisodg_2d_stage[c["shape"]](
                *args, *flags, lo, hi, want, m1[0], m1[1], m1[4], m1[5], m2[0], m2[1], m2[4], m2[5], d_max, d_dt)
        outs.append((host(d_out), host(d_max)[0]))
    (a, amax), (b, bmax) = outs
    assert np.isfinite(a[1:-1, 1:-1]).all()
    assert np.array_equal(a, b)
    assert amax == bmax


def test_dg_full_size_properties(hip_lib):
    """
    2048^2 zones (too slow for the CPU oracle): a uniform state at rest without masses is a fixed
    point of the limiter and (zone averages: to the last bit) of the update, and the update of a smooth state is invariant,
    bit for bit, under the split into row ranges.
    """
    import torch

    n = 2048
    extent = (-8.0, 8.0, -8.0, 8.0)
    c = cases.dg_case(ni=1, nj=1, masses=False, buffer_on=False, nu=0.01, eos=1, extent=extent, dt=1e-4)
    w = torch.zeros((n + 2, n + 2, 3, 3, 3), dtype=torch.float64, device="cuda")
    w[..., 0, 0, 0] = 1.25
    out = torch.full_like(w, -1.0)
    hip_lib.sfhip_cbdisodg_2d_limit[(n, n)](w, out, 1, 1, 1, 0, n)
    assert torch.equal(out[1:-1, 1:-1], w[1:-1, 1:-1])
    out.fill_(-1.0)
    zeros = (0.0,) * 8
    dmax = torch.zeros(1, dtype=torch.float64, device="cuda")
    base = cases.dg_advance_args(c, w, w, out)
    hip_lib.sfhip_cbdisodg_2d_stage[(n, n)](*base, 1, 1, 1, 0, n, 2, *zeros, dmax, None)
    used = [(m, k) for m in range(3) for k in range(3) if m + k < 3]
    for m, k in used:  # averages to the last bit; the pressure's volume and surface integrals cancel to rounding
        diff = (out[1:-1, 1:-1, :, m, k] - w[1:-1, 1:-1, :, m, k]).abs().max().item()
        assert (diff == 0.0) if m + k == 0 else (diff < 1e-15), (m, k, diff)
    assert float(dmax[0]) == np.sqrt(c["eos"][0])

    x = torch.linspace(-1.0, 1.0, n + 2, dtype=torch.float64, device="cuda")
    w[..., 0, 0, 0] = 1.0 + 0.2 * torch.sin(3.0 * x)[:, None] * torch.cos(2.0 * x)[None, :]
    w[..., 1, 0, 0] = 0.3 * torch.cos(2.0 * x)[:, None] * w[..., 0, 0, 0]
    w[..., 2, 0, 0] = -0.2 * torch.sin(4.0 * x)[None, :] * w[..., 0, 0, 0]
    w[..., 0, 1, 0] = 1e-3 * torch.cos(3.0 * x)[:, None]
    w[..., 0, 0, 1] = -1e-3 * torch.sin(2.0 * x)[None, :]
    whole, parts = torch.zeros_like(w), torch.zeros_like(w)
    hip_lib.sfhip_cbdisodg_2d_stage[(n, n)](*cases.dg_advance_args(c, w, w, whole), 1, 1, 1, 0, n, 0, *zeros, dmax, None)
    for lo, hi in ((0, 1), (1, 1000), (1000, n - 1), (n - 1, n)):
        hip_lib.sfhip_cbdisodg_2d_stage[(n, n)](*cases.dg_advance_args(c, w, w, parts), 1, 1, 1, lo, hi, 0, *zeros,
                                                dmax, None)
    assert torch.equal(whole, parts)
    assert not torch.equal(whole[1:-1, 1:-1], w[1:-1, 1:-1]) and bool(torch.isfinite(whole).all())
