#!/usr/bin/env python3
"""
Kernel micro-bench: times the fused stage kernel variants in isolation on one patch.

    python tools/kbench.py [--solver iso|gam] [--resolution 4096] [--reps 20]

For each variant (a = 0 without u^n; a = 1/2 with u^n, with and without the
wavespeed epilogue) it launches the kernel `reps` times through the product
path (Patch.launch_stage -> Library -> libsfhip.so), bracketed by HIP events
on the launch stream, and prints the mean duration and algorithmic GB/s.
Development tool: numbers quoted in DESIGN.md come from bench.py.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solver", default="iso", choices=["iso", "gam"])
    ap.add_argument("--resolution", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--patches", type=int, default=1,
                    help="cut the mesh into this many slab patches (all on this GPU) and time patch 0: the launch "
                         "shape one rank of a multi-GPU run sees")
    ap.add_argument("--sustained", type=int, default=0,
                    help="also time N back-to-back launches of each variant without host synchronisation in "
                         "between (the clocks the bench sees) and an alternation of the two RK2 stage variants")
    args = ap.parse_args()

    import torch
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk", eos="isothermal" if args.solver == "iso" else "gamma-law")
    mesh = setup.mesh(args.resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, mode="hip",
                         num_patches=args.patches)
    cfl = solver.recommended_cfl
    for _ in range(5):  # leave the analytic initial state: a developed flow is what the bench times
        solver.step(cfl)
    dt = mesh.min_spacing() / solver.maximum_wavespeed() * cfl
    p = solver.patches[0]
    ni, nj = p.shape
    masses = solver.point_masses(p.time)
    ncons = p.num_fields
    zones = ni * nj
    floor = solver._wavespeed_floor

    variants = [
        ("a=0   no u^n, no wavespeed", 0.0, None, 0, 2 * 8 * ncons),
        ("a=1/2 u^n,    no wavespeed", 0.5, p.primitive2, 0, 3 * 8 * ncons),
        ("a=1/2 u^n,    wavespeed   ", 0.5, p.primitive2, 2, 3 * 8 * ncons),
        ("a=0   no u^n, wavespeed   ", 0.0, None, 2, 2 * 8 * ncons),
        ("a=0   no u^n, no wavespeed", 0.0, None, 0, 2 * 8 * ncons),  # again: the first line warms the clocks
        # u^n read from the array the stencil reads (rows a wave loaded three iterations earlier: cache hits).
        # Same instructions as the u^n variant without its HBM traffic: what that traffic costs.
        ("a=1/2 u^n = rd (cached)     ", 0.5, p.primitive1, 0, 2 * 8 * ncons),
    ]
    scratch = torch.empty_like(p.primitive1)
    for name, a, un, ws, bytes_per_zone in variants:
        times = []
        for rep in range(args.reps + 3):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            fire = p.launch_stage(a, dt, p.primitive1, un, scratch, masses, masses, 0, ni, ws, floor, defer=True)
            e0.record()
            fire()
            e1.record()
            e1.synchronize()
            if rep >= 3:
                times.append(e0.elapsed_time(e1))
        ms = sum(times) / len(times)
        print(f"{args.solver} {ni}x{nj} {name}: {ms:.4f} ms  min {min(times):.4f}  "
              f"{zones * bytes_per_zone / (ms * 1e-3) / 1e9:7.1f} GB/s algorithmic", flush=True)

    if args.sustained:
        def burst(seq, label):
            fires = [p.launch_stage(a, dt, p.primitive1, un, scratch, masses, masses, 0, ni, ws, floor, defer=True)
                     for (_, a, un, ws, _) in seq]
            for _ in range(2):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.sustained):
                    for fire in fires:
                        fire()
                e1.record()
                e1.synchronize()
            print(f"sustained {label}: {e0.elapsed_time(e1) / args.sustained:.4f} ms per round of {len(seq)} launch(es)", flush=True)
        burst([variants[0]], "a=0 no wavespeed")
        burst([variants[2]], "a=1/2 u^n wavespeed")
        burst([variants[0], variants[2]], "RK2 pair")


if __name__ == "__main__":
    main()
