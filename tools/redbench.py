#!/usr/bin/env python3
"""Times Solver.reductions() (fused on-device diagnostics, one pass per 8 diagnostics) at 4096^2.
usage: python tools/redbench.py [--solver iso|gam]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", default="iso")
    ap.add_argument("--resolution", type=int, default=4096)
    args = ap.parse_args()
    import torch
    from sailfish_amd.physics.circumbinary import Diagnostic
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk", eos="isothermal" if args.solver == "iso" else "gamma-law")
    mesh = setup.mesh(args.resolution)
    diagnostics = [Diagnostic("mass"), Diagnostic("mdot", accretion=True, which_mass="both"),
                   Diagnostic("torque", gravity=True, which_mass="both"), Diagnostic("angular_momentum"),
                   Diagnostic("power", gravity=True, which_mass=1), Diagnostic("sigma_m1"),
                   Diagnostic("eccentricity_vector"), Diagnostic("fx", gravity=True, which_mass=2)]
    physics = dict(setup.physics, diagnostics=diagnostics)
    s = make_solver(setup.solver, physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    for _ in range(3):
        s.step(s.recommended_cfl)
    s.reductions()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        s.reductions()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    nb = mesh.num_total_zones * 8 * s.num_cons
    print(f"{args.solver} {args.resolution}^2: reductions() of {len(diagnostics)} diagnostics {ms:.3f} ms per call "
          f"({nb / (ms * 1e-3) / 1e9:.0f} GB/s of primitive array)")


if __name__ == "__main__":
    main()
