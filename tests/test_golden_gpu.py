"""
GPU parity against the reference's own runs (tests/golden/*.npz): the HIP solver,
started from the fixture's initial data, must land within 1e-12 relative per zone
of the reference's final state after the same number of steps; its first dt is
bit-identical and later ones agree to 1e-13 (see test_hip_solver.run_pair for why
not bitwise).
"""
import os

import numpy as np
import pytest

from sailfish_amd.driver import timestep
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
from test_golden_cpu import CASES, load, start_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_solver_matches_reference_run(name, fused):
    setup_name, model, options = CASES[name]
    g = load(name)
    setup = make_setup(setup_name, **model)
    mesh = setup.mesh(int(g["resolution"]))
    s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=start_of(g)[0],
                    solution=g["initial"], num_patches=int(g["num_patches"]), mode="hip", fused=fused)
    cfl = float(g["cfl"])
    for k, (dt_ref, t_ref) in enumerate(zip(g["dt"], g["time"])):
        dt = timestep(mesh, s, cfl)
        if k == 0:
            assert dt == dt_ref
        else:
            assert abs(dt - dt_ref) <= 1e-13 * dt_ref
        s.advance(dt)
        assert abs(s.time - t_ref) <= 1e-13 * t_ref
    from test_hip_solver import check

    check(s.solution, g["final"])  # 1e-12 per zone, and the figures go to gpurun_out/parity_figures.txt
