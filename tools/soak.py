import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
for model, n, steps in (({}, 1024, 20000), (dict(eos="gamma-law"), 512, 20000), ({}, 256, 50000)):
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(n)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    cfl = s.recommended_cfl
    t0 = time.perf_counter()
    for k in range(steps):
        dt = s.step(cfl)
        if k % 5000 == 0:
            print(model, n, k, dt, s.time, s.wavespeed_redos, flush=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    sol = s.solution
    print(f"{model} {n}^2: {steps} steps in {el:.1f} s = {el / steps * 1e6:.1f} us/step, t = {s.time:.4f}, redos {s.wavespeed_redos}, "
          f"finite {np.isfinite(sol).all()}, mass {sol[..., 0].sum() * mesh.dx * mesh.dy:.6f}", flush=True)
