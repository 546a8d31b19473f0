import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sailfish_amd.setups import make_setup
from sailfish_amd.solvers import make_solver
setup = make_setup("circumbinary-disk", eos="gamma-law")
mesh = setup.mesh(384)
s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
cfl = s.recommended_cfl
prev = None
for k in range(3000):
    if k >= 1000:
        prev = (s.solution.copy(), s.time)
    dt = s.step(cfl)
    if k >= 1000:
        p = s.solution
        vmax = np.abs(p[..., 1:3]).max()
        if vmax > 50 or not np.isfinite(p).all():
            i, j = np.unravel_index(np.argmax(np.abs(p[..., 1:3]).max(-1)), p.shape[:2])
            print("step", k, "t", s.time, "dt", dt, "worst zone", i, j, "vmax", vmax)
            np.set_printoptions(precision=17, linewidth=200)
            print("before:\n", prev[0][i-3:i+4, j-3:j+4])
            print("after:\n", p[i-1:i+2, j-1:j+2])
            np.savez("gpurun_out/gamcatch.npz", before=prev[0][i-8:i+9, j-8:j+9], after=p[i-8:i+9, j-8:j+9], t=prev[1], dt=dt, i=i, j=j)
            break
