"""
GPU parity of the srhd_2d entry points (sailfish_amd/csrc/sf_srhd.hpp) against the CPU
oracle (oracle/sf_oracle_srhd.c, itself bit-identical to the compiled reference).

Bars: the device kernels perform the reference's IEEE operations in the reference's
order with the polar trigonometry from a host libm table, so everything that involves
no device transcendental must be BIT-EXACT (p2c, c2p, wavespeed, advance without jet);
the jet inflow row (exp/pow on the device) is held to 1e-12 relative per zone.
"""
import ctypes

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
    lib = oracle.oracle()
    lib.dll.sfo_srhd_2d_set_macros.argtypes = [ctypes.c_double] * 2
    return lib


@pytest.fixture(params=[(1.5, 1e6), (2.0, 100.0)], ids=["theta1.5", "theta2.0-mach100"])
def macros(request, orc, hip_lib):
    orc.dll.sfo_srhd_2d_set_macros(*request.param)
    hip_lib.sfhip_srhd_2d_set_macros[1](*request.param)
    yield request.param
    orc.dll.sfo_srhd_2d_set_macros(1.5, 1e6)
    hip_lib.sfhip_srhd_2d_set_macros[1](1.5, 1e6)


@pytest.mark.parametrize("kw", cases.srhd_branch_matrix(), ids=IDS)
def test_srhd_advance(kw, macros, orc, hip_lib):
    c = cases.srhd_case(seed=13, **kw)
    cons_rd = cases.srhd_conserved(orc, c)
    cons_rk = cases.srhd_conserved(orc, c, prim=c["prim"] * 1.01)
    expect = np.full_like(c["prim"], -7.0)
    orc.call("srhd_2d_advance_rk", c["shape"], *cases.srhd_advance_args(c, cons_rk, c["prim"], cons_rd, expect))
    d_out = dev(np.full_like(c["prim"], -7.0))
    args = cases.srhd_advance_args(c, dev(cons_rk), dev(c["prim"]), dev(cons_rd), d_out)
    hip_lib.srhd_2d_advance_rk[c["shape"]](dev(c["faces"]), *args[1:])
    got = host(d_out)
    jet_on = c["jet"][0] > 0.0
    body = slice(3, None) if jet_on else slice(None)
    assert np.array_equal(got[body], expect[body])  # bit-exact away from the jet row
    if jet_on:
        scale = np.abs(expect[2]).max(axis=-1, keepdims=True)
        assert (np.abs(got[2] - expect[2]) <= TOL * scale).all()


@pytest.mark.parametrize("cold", [False, True])
@pytest.mark.parametrize("adot", [0.0, 0.15])
def test_srhd_p2c_c2p_wavespeed_bit_exact(macros, cold, adot, orc, hip_lib):
    c = cases.srhd_case(seed=21, ni=30, nj=17, adot=adot, time=1.4, cold=cold)
    cons = cases.srhd_conserved(orc, c)
    inner = np.ascontiguousarray(c["prim"][2:-2])
    d_cons = dev(np.zeros_like(inner))
    d_faces = dev(c["faces"])
    hip_lib.srhd_2d_primitive_to_conserved[c["shape"]](d_faces, dev(inner), d_cons, c["polar_extent"], c["scale_factor"])
    assert np.array_equal(host(d_cons), cons[2:-2])

    rng = np.random.default_rng(5)
    guess = c["prim"].copy()
    guess[..., 3] *= rng.uniform(0.7, 1.4, guess.shape[:2])
    p, u2, w = guess.copy(), np.full_like(cons, -3.0), np.full(c["shape"], -1.0)
    orc.call("srhd_2d_conserved_to_primitive", c["shape"], c["faces"], cons, u2, p, c["polar_extent"], c["scale_factor"])
    orc.call("srhd_2d_max_wavespeeds", c["shape"], c["faces"], p, w, c["adot"])
    d_p, d_u2, d_w = dev(guess), dev(np.full_like(cons, -3.0)), dev(np.full(c["shape"], -1.0))
    hip_lib.srhd_2d_conserved_to_primitive[c["shape"]](d_faces, dev(cons), d_u2, d_p, c["polar_extent"], c["scale_factor"])
    hip_lib.srhd_2d_max_wavespeeds[c["shape"]](d_faces, d_p, d_w, c["adot"])
    assert np.array_equal(host(d_p), p)
    assert np.array_equal(host(d_u2), u2)
    assert np.array_equal(host(d_w), w)
    count = dev(np.array([-1.0]))
    hip_lib.sfhip_srhd_2d_failures[1](count)
    assert host(count)[0] == 0.0 and orc.dll.sfo_srhd_2d_failures() == 0


def test_srhd_c2p_counts_failures_instead_of_exiting(orc, hip_lib):
    """Where the reference prints and exit(1)s (srhd_2d.c:164-184) the device path counts the zone."""
    c = cases.srhd_case(seed=2, ni=6, nj=5)
    cons = cases.srhd_conserved(orc, c)
    cons[3, 2, 3] = -1.0  # negative total energy in one zone
    cons[4, 1, :] = np.nan
    d_p = dev(c["prim"])
    hip_lib.srhd_2d_conserved_to_primitive[c["shape"]](dev(c["faces"]), dev(cons), dev(np.zeros_like(cons)), d_p,
                                                        c["polar_extent"], c["scale_factor"])
    count = dev(np.array([-1.0]))
    hip_lib.sfhip_srhd_2d_failures[1](count)
    assert host(count)[0] == 2.0
    hip_lib.sfhip_srhd_2d_failures[1](count)
    assert host(count)[0] == 0.0  # reading clears the counter


def test_srhd_bad_arguments_are_refused(hip_lib):
    from sailfish_amd.kernel.library import KernelError

    c = cases.srhd_case(ni=4, nj=4)
    d = dev(c["prim"])
    with pytest.raises(KernelError):
        hip_lib.srhd_2d_advance_rk[c["shape"]](dev(c["faces"]), d, d, d, d, np.pi, 1.0, 0.0, 1.0, 0.0, 1e-3,
                                               0.0, 0.0, 0.0, 0.0, 0)  # wr aliases the inputs
    with pytest.raises(KernelError):
        hip_lib.sfhip_srhd_2d_set_macros[1](3.0, 1e6)
