import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
setup = make_setup("circumbinary-disk", eos="gamma-law")
mesh = setup.mesh(384)
s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
cfl = s.recommended_cfl
for k in range(6000):
    dt = s.step(cfl)
    if k % 500 == 499:
        p = s.solution
        print(os.environ.get("SFHIP_LIBRARY","tree"), k+1, "t=%.5f dt=%.3e" % (s.time, dt), "min rho %.3e min p %.3e max|v| %.3e" % (p[...,0].min(), p[...,3].min(), np.abs(p[...,1:3]).max()), flush=True)
