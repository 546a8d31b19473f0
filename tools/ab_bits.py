#!/usr/bin/env python3
"""
Same bits from two builds of the library?  Runs 30 steps of the iso and the gamma-law disk at 256^2 (and 60 of the DG
disk at 192^2) with the library
SFHIP_LIBRARY names (default: the tree's) and writes the final states to the given .npz; run it once per build and
compare with `--compare a.npz b.npz` (development tool for changes that claim to be bit-neutral).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        print(k, "identical" if same else f"DIFFERENT: max abs {np.abs(a[k] - b[k]).max():.3e}")
    sys.exit(0)

from sailfish_amd.setups import make_setup  # noqa: E402
from sailfish_amd.solvers import make_solver  # noqa: E402

out = {}
for name, model in (("iso", {}), ("gam", dict(eos="gamma-law"))):
    setup = make_setup("circumbinary-disk", **model)
    mesh = setup.mesh(256)
    s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    dts = [s.step(s.recommended_cfl) for _ in range(30)]
    out[name] = s.solution
    out[name + "_dt"] = np.array(dts)
# cbdisodg_2d: 60 steps of the kitp disk at 192^2 (for SFHIP_DG_ZONE_KERNEL=1 against the marching kernel)
setup = make_setup("kitp-code-comparison", use_dg=True)
s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=setup.mesh(192), time=0.0, mode="hip")
dts = [s.step(s.recommended_cfl) for _ in range(60)]
out["dg"] = s.solution
out["dg_dt"] = np.array(dts)
np.savez(sys.argv[1], **out)
