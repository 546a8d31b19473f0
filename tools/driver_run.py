#!/usr/bin/env python3
"""
The reference's own figure of merit, printed as its driver prints it: `sailfish_amd.driver.run` on the
circumbinary disk, Mzps per fold of 10 iterations (driver.py:506-517).   usage: tools/driver_run.py [resolution] [iterations]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sailfish_amd.driver import run  # noqa: E402

resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iterations = int(sys.argv[2]) if len(sys.argv) > 2 else 300
solver, reports = run("circumbinary-disk", resolution=resolution, max_iterations=iterations, fold=10)
for r in reports[:3] + reports[-5:]:
    print(f"[{r.iteration:04d}] t={r.time:.5f} dt={r.dt:.3e} Mzps={r.mzps:.1f}")
tail = reports[len(reports) // 2:]
print(f"mean of the last {len(tail)} folds: {sum(r.mzps for r in tail) / len(tail):.1f} Mzps")
