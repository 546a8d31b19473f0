#!/usr/bin/env python3
"""
Host-side cost of one driver iteration: cProfile of Solver.step() on a mesh small enough (256^2)
that the GPU is never the bottleneck.  What this shows bounds the step rate the host can feed,
which is what limits weak scaling once per-GPU kernels take well under a millisecond.
usage: python tools/hostprof.py [--solver iso|gam] [--steps 300]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", default="iso")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--resolution", type=int, default=256)
    args = ap.parse_args()
    import torch
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk", eos="isothermal" if args.solver == "iso" else "gamma-law")
    mesh = setup.mesh(args.resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip")
    cfl = solver.recommended_cfl
    for _ in range(20):
        solver.step(cfl)
    torch.cuda.synchronize()
    plan = getattr(solver.patches[0], "plan", None)
    if plan is not None:
        plan.step_timing()  # reset
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.step(cfl)
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / args.steps * 1e6:.1f} us per step (host-bound at {args.resolution}^2)")
    if plan is not None:
        a, b, c, d, n = plan.step_timing()
        if n:
            print(f"inside sfhip_plan_step_rk2 ({int(n)} calls): stage-0 launch {a:.1f} us, wait for the hand-over {b:.1f} us, "
                  f"dt + orbit {c:.1f} us, stage-1 launch {d:.1f} us")
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(args.steps):
        solver.step(cfl)
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
