#!/usr/bin/env python3
"""
Long run of the gamma-law circumbinary disk at 512^2 on the GPU (development tool): 16 000 steps, then a state
every 100 steps (the last three kept under gpurun_out/) until the solution blows up.  The reference scheme itself
goes unstable there at t = 20.4135 (velocity ceiling reached next to a sink, dt -> 3e-19); tools/longrun_cpu_check.py
continues the last saved state with the compiled reference on the CPU and shows the same end.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
setup = make_setup("circumbinary-disk", eos="gamma-law")
mesh = setup.mesh(512)
s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
cfl = s.recommended_cfl
os.makedirs("gpurun_out", exist_ok=True)
k = 0
saved = []
while k < 21000:
    n = 16000 if k == 0 else 100
    for _ in range(n):
        dt = s.step(cfl); k += 1
    sol = s.solution
    bad = not np.isfinite(sol).all() or sol[..., 3].max() > 1e3 or np.abs(sol[..., 1:3]).max() > 1e3
    print(k, f"t={s.time:.5f} dt={dt:.3e} max sigma={sol[...,0].max():.3e} min sigma={sol[...,0].min():.3e} max p={sol[...,3].max():.3e} min p={sol[...,3].min():.3e} max |v|={np.abs(sol[...,1:3]).max():.3e} redos={s.wavespeed_redos}", flush=True)
    if bad:
        print("blow-up detected at step", k)
        break
    path = f"gpurun_out/gam_state_{k}.npz"
    np.savez(path, solution=sol, time=s.time, step=k)
    saved.append(path)
    if len(saved) > 3:
        os.remove(saved.pop(0))
