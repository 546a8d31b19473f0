"""
BASELINE.json configs[2] and configs[3] at their REAL per-rank shapes on one GPU:

  * cbdiso_2d 16384^2 cut into 8 slab patches of 2048 x 16384 zones (12.9 GB of state),
  * cbdgam_2d  8192^2 cut into 8 slab patches of 1024 x  8192 zones.

The oracle cannot run these sizes in seconds, so the full mesh is checked through size-independent
properties -- the 8-patch run equals the 1-patch run bit for bit (state and dt sequence; power-of-two
meshes are decomposition invariant in the reference), reductions identities --
and the oracle is run on WINDOWS of the same mesh (a patch seam, the binary, a domain corner, the last
columns of the 16 388-column rows): the reference's RK2 step of a window, started from the window's
initial data, equals the device's step of the whole mesh inside the window's domain of dependence.
"""
import numpy as np
import pytest

import cases
import ref_solver
from sailfish_amd.physics.circumbinary import Diagnostic
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
from sailfish_amd.subdivide import subdivide

TOL = 1e-12
MARGIN = 6  # zones of a window that its artificial outflow edge can reach in one RK2 step (2 stages x reach 2), plus 2


def window_expectation(setup, mesh, model_time, dt, rows, cols):
    """The reference's step (CPU oracle, reference order of kernels) of zones rows x cols started from initial data."""
    sub = mesh.sub_mesh(rows, cols)
    nq = 3 if setup.solver == "cbdiso_2d" else 4
    full_rows = setup.primitive_array(model_time, mesh, rows[0], rows[1], nq)
    cpu = ref_solver.OracleSolver(setup, sub, time=model_time, solution=np.ascontiguousarray(full_rows[:, cols[0]:cols[1]]))
    # the buffer is a property of the whole domain, not of the window (cbdiso_2d.py:325-337)
    if cpu.physics.buffer_is_enabled:
        sample = [0.0] * nq
        setup.primitive(model_time, [mesh.x1 - cpu.physics.buffer_onset_width, 0.0], sample)
        cpu.buffer_outer_radius = mesh.x1
        cpu.buffer_surface_density = sample[0]
        cpu.buffer_surface_pressure = sample[3] if nq == 4 else 0.0
    cpu.advance(dt)
    return cpu.solution


def windows_of(n, num_patches):
    """(rows, cols) index ranges: a seam between two patches, the binary, two corners, the row ends."""
    seam = subdivide(n, num_patches)[num_patches // 2][0]
    h = n // 2
    w = min(40, n // 16)
    return {
        "patch seam": ((seam - w, seam + w), (h - 3 * w, h + 3 * w)),
        "second seam, row end": ((subdivide(n, num_patches)[1][0] - w, subdivide(n, num_patches)[1][0] + w), (n - 5 * w, n)),
        "binary": ((h - 2 * w, h + 2 * w), (h - 2 * w, h + 2 * w)),
        "lower corner": ((0, 2 * w), (0, 2 * w)),
        "upper corner": ((n - 2 * w, n), (n - 2 * w, n)),
    }


def interior_of(window, n):
    """Slices of the window (in window coordinates) outside the reach of its artificial edges."""
    out = []
    for a, b in window:
        lo = 0 if a == 0 else MARGIN          # a real domain edge is a real outflow edge: exact
        hi = (b - a) if b == n else (b - a) - MARGIN
        out.append(slice(lo, hi))
    return tuple(out)


def device_block(solver, rows, cols):
    """Zones rows x cols of the device state as a host array, without gathering the whole mesh."""
    parts = []
    for p in solver.patches:
        a, b = p.index_range
        lo, hi = max(a, rows[0]), min(b, rows[1])
        if lo < hi:
            parts.append(p.primitive1[2 + lo - a:2 + hi - a, 2 + cols[0]:2 + cols[1]].cpu().numpy())
    return np.concatenate(parts, axis=0)


def states_equal_on_device(one, many):
    """Bitwise comparison of a 1-patch and a multi-patch solver's states where they live."""
    import torch

    whole = one.patches[0].primitive1
    for p in many.patches:
        a, b = p.index_range
        if not torch.equal(whole[2 + a:2 + b, 2:-2], p.primitive1[2:-2, 2:-2]):
            return False
    return True


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n", [("iso", 16384), ("gam", 8192)])
def test_config_at_its_per_rank_shape(kind, n):
    check_config(kind, n, dict(mode="hip"))


@pytest.mark.parametrize("kind", ["iso", "gam"])
def test_window_checks_on_the_cpu(kind):
    """The same checks at 256^2 with the oracle behind the solver: the window machinery itself is exact there
    (what the windows see of the full run is bit-identical up to the coordinates' last place)."""
    from oracle_backend import OracleBackend

    check_config(kind, 256, dict(backend=OracleBackend(), distributed=False))


def check_config(kind, n, how):
    model = {} if kind == "iso" else dict(eos="gamma-law")
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(n)
    diagnostics = [
        Diagnostic("mass"), Diagnostic("mass", radial_cut=(0.0, 5.0)), Diagnostic("mass", radial_cut=(5.0, 1e9)),
        Diagnostic("mdot", accretion=True, which_mass="both"), Diagnostic("mdot", accretion=True, which_mass=1),
        Diagnostic("mdot", accretion=True, which_mass=2), Diagnostic("angular_momentum"),
    ]
    physics = dict(setup.physics, diagnostics=diagnostics)
    kw = dict(setup=setup, mesh=mesh, time=0.0, **how)
    many = make_solver(setup.solver, physics, {}, num_patches=8, **kw)
    assert [p.shape for p in many.patches] == [(n // 8, n)] * 8  # 2048 x 16384 / 1024 x 8192: the per-rank shape
    cfl = many.recommended_cfl

    # step 1 against the reference's step on windows of the mesh
    dt = many.step(cfl)
    worst = 0.0
    for name, (rows, cols) in windows_of(n, 8).items():
        expect = window_expectation(setup, mesh, 0.0, dt, rows, cols)
        got = device_block(many, rows, cols)
        keep = interior_of((rows, cols), n)
        err = cases.zone_relative_error(got[keep], expect[keep], interior=False)
        worst = max(worst, err)
        assert err <= TOL, (name, err)
    print(f"{setup.solver} {n}^2 in 8 patches: worst window error after one RK2 step {worst:.2e}")

    dts = [dt] + [many.step(cfl) for _ in range(2)]
    one = make_solver(setup.solver, physics, {}, num_patches=1, **kw)
    assert [one.step(cfl) for _ in range(3)] == dts  # dt sequence: bit for bit
    assert states_equal_on_device(one, many)
    assert one.time == many.time

    r1, r8 = np.array(one.reductions()), np.array(many.reductions())
    assert np.allclose(r1, r8, rtol=1e-11, atol=1e-13)
    m, m_in, m_out, md, md1, md2, lz = r8
    assert abs(m - (m_in + m_out)) <= 1e-12 * m
    assert abs(md - (md1 + md2)) <= 1e-12 * (abs(md1) + abs(md2))
    assert md1 < 0.0 and md2 < 0.0 and lz > 0.0 and m > 0.0
    if kind == "iso":
        assert abs(m - 24.0**2) < 0.01 * 24.0**2  # surface density 1 on [-12, 12]^2
