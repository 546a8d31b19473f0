"""
The entry point that allocates nothing: sfhip_cbdiso_2d_reductions_noalloc (include/sfhip.h) against the
existing sfhip_cbdiso_2d_reductions, through the C ABI.  Same kernels and launch geometry, block partials in
the library's built-in device array instead of a caller's workspace: the sums are bit-identical (the bar of
the existing entry point's own determinism test).  Two calls in a row, the second on a smaller patch (the
array keeps the first call's partials; nothing may leak from them), then oversize shapes are refused.

Collected last (file name): a failure here must not hide the other GPU tests behind `pytest -x`.
"""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu

SPEC = np.array([
    # code, which, gravity, accretion, has_cut, r0, r1, unused
    [15, 0, 0, 0, 0, 0, 0, 0],        # mass
    [15, 0, 0, 0, 1, 0.5, 1.5, 0],    # mass in an annulus
    [0, 3, 0, 1, 0, 0, 0, 0],         # mdot, both masses, accretion
    [10, 1, 1, 0, 0, 0, 0, 0],        # torque of mass 1, gravity
    [11, 2, 1, 0, 0, 0, 0, 0],        # power on mass 2, gravity
    [12, 0, 0, 0, 0, 0, 0, 0],        # sigma_m1 (complex)
    [13, 0, 0, 0, 1, 0.2, 2.0, 0],    # eccentricity vector (complex), cut
    [14, 0, 0, 0, 0, 0, 0, 0],        # angular momentum
], dtype=float)


def both_ways(hip_lib, seed, ni, nj):
    import torch

    c = cases.iso_case(seed=seed, ni=ni, nj=nj)
    xl, xr, yl, yr = c["extent"]
    dx, dy = (xr - xl) / ni, (yr - yl) / nj
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    prim = dev(c["prim"])
    x = dev(np.linspace(xl + 0.5 * dx, xr - 0.5 * dx, ni))
    y = dev(np.linspace(yl + 0.5 * dy, yr - 0.5 * dy, nj))
    spec = dev(SPEC)
    nd = len(SPEC)
    n = hip_lib.reductions_workspace_len(ni, nj)
    workspace = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
    expect = torch.full((2 * nd,), -1.0, dtype=torch.float64, device="cuda")
    got = torch.full((2 * nd,), -2.0, dtype=torch.float64, device="cuda")
    hip_lib.sfhip_cbdiso_2d_reductions[(ni, nj)](
        *c["extent"], *c["m1"], *c["m2"], prim, x, y, nd, spec, n, workspace, expect)
    hip_lib.sfhip_cbdiso_2d_reductions_noalloc[(ni, nj)](
        *c["extent"], *c["m1"], *c["m2"], prim, x, y, nd, spec, got)
    return got.cpu().numpy(), expect.cpu().numpy()


def test_noalloc_reductions_match_the_workspace_entry_point(hip_lib):
    got, expect = both_ways(hip_lib, 5, 300, 700)  # 3 x 300 blocks: most of the scratch array in use
    assert np.isfinite(expect).all() and np.abs(expect).max() > 0.0
    assert np.array_equal(got, expect)
    got, expect = both_ways(hip_lib, 6, 38, 102)   # smaller, right behind it: the array holds the first call's partials
    assert np.isfinite(expect).all() and np.abs(expect).max() > 0.0
    assert np.array_equal(got, expect)
    # one zone, centred on the origin: sigma_m1 and the eccentricity vector are 0/0 there, in the reference too
    got, expect = both_ways(hip_lib, 7, 1, 1)
    assert np.isnan(expect).any() and np.array_equal(got, expect, equal_nan=True)


def test_noalloc_reductions_refuse_what_does_not_fit(hip_lib):
    import torch

    from sailfish_amd.kernel.library import KernelError

    one = torch.zeros(8, dtype=torch.float64, device="cuda")
    m = cases.NO_MASS
    hip_lib_nodebug = hip_lib  # debug validation would reject the dummy arrays before the library sees the shape
    debug = hip_lib_nodebug.debug
    hip_lib_nodebug.debug = False
    try:
        with pytest.raises(KernelError, match="scratch array"):
            hip_lib.sfhip_cbdiso_2d_reductions_noalloc[(4, 300000)](0.0, 1.0, 0.0, 1.0, *m, *m, one, one, one, 1, one, one)
    finally:
        hip_lib_nodebug.debug = debug
