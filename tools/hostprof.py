#!/usr/bin/env python3
"""
Host-side cost of one driver iteration: cProfile of Solver.step() on a mesh small enough (256^2)
that the GPU is never the bottleneck.  What this shows bounds the step rate the host can feed,
which is what limits weak scaling once per-GPU kernels take well under a millisecond.
usage: python tools/hostprof.py [--solver iso|gam] [--steps 300] [--resolution 256] [--patches P [--self-exchange]]

--patches P cuts the mesh into P slabs on this GPU (a step group: sfhip_step_group_rk2); with --self-exchange
their halo slabs travel through a one-rank RCCL communicator (grouped ncclSend / ncclRecv to the own rank on the
side stream), which is all of the multi-rank step a one-GPU box can run.  `--resolution 4096 --patches 8
--self-exchange` steps eight 512 x 4096 slabs -- the per-rank patch of BASELINE's 4096^2-over-8-GPUs line -- so the
host time per step and PATCH printed for it is what one rank of that run spends (an upper bound: it also posts
both sides of every exchange).
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
    ap.add_argument("--patches", type=int, default=1)
    ap.add_argument("--self-exchange", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    args = ap.parse_args()
    if args.self_exchange:
        os.environ["SFHIP_SELF_EXCHANGE"] = "1"
    import torch
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk", eos="isothermal" if args.solver == "iso" else "gamma-law")
    mesh = setup.mesh(args.resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip",
                         num_patches=args.patches, distributed=False)
    cfl = solver.recommended_cfl
    for _ in range(20):
        solver.step(cfl)
    torch.cuda.synchronize()
    plan = getattr(solver.patches[0], "plan", None)
    group = getattr(solver, "_group", None)
    if group is not None:
        plan = group  # same timing interface
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
            what = "sfhip_step_group_rk2" if group is not None else "sfhip_plan_step_rk2"
            print(f"inside {what} ({int(n)} calls): stage-0 launch {a:.1f} us, wait for the hand-over {b:.1f} us, "
                  f"dt + orbit {c:.1f} us, stage-1 launch {d:.1f} us")
            if group is not None:
                transport = "RCCL to the own rank" if args.self_exchange else "device slab copies"
                print(f"host time per step without the wait: {a + c + d:.1f} us for {args.patches} patches of "
                      f"{solver.patches[0].shape[0]} x {solver.patches[0].shape[1]} ({transport}) = "
                      f"{(a + c + d) / args.patches:.1f} us per patch")
    if args.no_profile:
        return
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(args.steps):
        solver.step(cfl)
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
