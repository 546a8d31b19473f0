#!/usr/bin/env python3
"""
Continues a state saved by tools/longrun_gpu.py with the reference's own C kernels (oracle/_ref, OpenMP) on the CPU,
dt recomputed every step as the driver does: is a blow-up on the GPU the scheme's or ours?
usage: python tools/longrun_cpu_check.py [gpurun_out/gam_state_<step>.npz]
"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle as test_oracle, ref_solver
from sailfish_amd.setups import make_setup
d = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "gam_state_18400.npz"))
sol, t0 = d["solution"], float(d["time"])
setup = make_setup("circumbinary-disk", eos="gamma-law")
mesh = setup.mesh(512)
lib = test_oracle.ref("cbdgam_2d", "omp")
print("reference lib:", lib.path if lib else None)
s = ref_solver.OracleSolver(setup, mesh, time=t0, lib=lib, solution=sol)
for k in range(1, 161):
    dt = mesh.min_spacing() / s.maximum_wavespeed() * 0.1
    s.advance(dt)
    if k % 10 == 0:
        u = s.solution
        print(k, f"t={s.time:.5f} dt={dt:.3e} max sigma={u[...,0].max():.3e} min sigma={u[...,0].min():.3e} max p={u[...,3].max():.3e} min p={u[...,3].min():.3e} max |v|={np.abs(u[...,1:3]).max():.3e}", flush=True)
        if not np.isfinite(u).all() or u[...,3].max() > 1e3:
            print("reference blows up too"); break
