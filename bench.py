#!/usr/bin/env python3
"""
bench.py -- zone-updates/sec of the cbdiso_2d hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is what the reference's driver does per iteration (driver.py:508-511):
dt from the max-wavespeed reduction, then one full RK2 ``solver.advance(dt)``.
Workload: BASELINE.json configs[1], the circumbinary-disk setup at 4096^2 fp64
on one MI355X (single patch).  For N > 1 the per-GPU work is held at 4096^2
zones (weak scaling): the square mesh grows to 4096*sqrt(N) per side and is cut
into N slab patches, one per rank, with halo slabs and the dt MAX all-reduce on
RCCL.  Initial data are the setup's analytic disk (no RNG, no files): resident
in HBM before the clock starts.

Output (rank 0): ONE JSON line with the driver's contract fields plus
  roofline     -- the stage kernel's achieved algorithmic GB/s against HBM peak,
                  launch durations measured with HIP events inside the timed region
  cpu_baseline -- the reference's own C/OpenMP kernels (oracle/_ref, built from
                  /root/reference by oracle/build_ref.py) or, where absent, our
                  C port (oracle/sf_oracle.c), driven in the reference's order
                  on a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import json
import math
import os
import sys
import time

# the host driver of this pool only supports dmabuf IPC: RCCL between processes needs this (already exported
# on the boxes; kept here so that the multi-rank launch does not depend on the caller's environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# Algorithmic bytes per zone: a stage reads p and writes p' (NCONS fp64 each); a stage
# with a != 0 also reads the u^n source at the zone centre.  RK2 zone-update = one of each.
NCONS = {"iso": 3, "gam": 4, "srhd": 4, "dg": 3}
# srhd_2d, fused stages (sfhip_srhd_2d_stage): a stage reads the primitives and the conserved state (and u^n
# when rk_param != 0) and writes the updated state, its floored copy and its primitives: 5 or 6 arrays of 32 B
# per zone; RK2 zone-update = 11 x 32 B.  (The reference's sequence moves 18 arrays: 576 B.)
SRHD_BYTES_PER_ZONE_UPDATE = 11 * 32.0
# cbdisodg_2d: a zone is 3 fields x 6 Legendre weights (stored 3 x 9).  Per RK stage the limiter copies all 27
# doubles (216 B in, 216 B out, as the reference's kernel does) and the update reads u^n and w and writes w'
# (18 used doubles each, 432 B): 864 B per stage, RK2 zone-update = 1728 B.
DG_BYTES_PER_ZONE_UPDATE = 2 * 864.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--resolution", type=int, default=None, help="zones per side (default 4096*sqrt(gpus))")
    ap.add_argument("--solver", default="iso", choices=["iso", "gam", "srhd", "dg"])
    ap.add_argument("--unfused", action="store_true", help="drive the reference-named kernels in reference order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time for the baseline sample")
    ap.add_argument("--sync-dt", action="store_true", help="dt = dx/maximum_wavespeed()*cfl; advance(dt) as two blocking calls")
    ap.add_argument("--no-events", action="store_true", help="do not bracket launches with HIP events")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket the kernel launches of every Nth timed step with HIP events (the brackets cost "
                         "~3.5 us per launch: 2.4 %% of the step at stride 1, measured)")
    ap.add_argument("--strong", action="store_true",
                    help="hold the MESH at 4096^2 as N grows (strong scaling) instead of the zones per GPU")
    return ap.parse_args()


def recorded_traffic(solver, resolution, world):
    """HBM bytes per launch of the dominant kernel from the PMC passes of tools/profile.sh (profiles/traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, in KiB, as the MI355X guide prescribes), or None for an unprofiled configuration."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        return json.load(f).get(f"{solver}_{resolution}_{world}")


def default_resolution(gpus):
    n = 4096.0 * math.sqrt(gpus)
    return int(n // 8) * 8


def cpu_baseline(setup_name, setup_kwargs, resolution, target_seconds):
    """Time the reference's CPU path (or our port of it) on a few steps of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as test_oracle
    import ref_solver
    from sailfish_amd.setups import make_setup

    setup = make_setup(setup_name, **setup_kwargs)
    mesh = setup.mesh(resolution)
    lib = test_oracle.ref(setup.solver, "omp")
    kind = "reference"
    if lib is None:
        lib = test_oracle.oracle()
        kind = "port"
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(os.sched_getaffinity(0))
    if setup.solver in ("srhd_2d", "cbdisodg_2d"):
        # the reference's solver loop is our host mirror of it (pinned bitwise by tests/test_srhd_solver.py and
        # tests/test_dg_solver.py) on the reference's own C kernels
        from oracle_backend import OracleBackend
        from sailfish_amd.solvers import make_solver

        extra = dict(fused=False) if setup.solver == "cbdisodg_2d" else {}
        solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=setup.start_time,
                             backend=OracleBackend(lib), distributed=False, **extra)
        cfl = solver.recommended_cfl
    else:
        solver = ref_solver.OracleSolver(setup, mesh, time=0.0, lib=lib)
        cfl = 0.3 if setup.solver == "cbdiso_2d" else 0.1

    def step():
        dt = mesh.min_spacing(solver.time) / solver.maximum_wavespeed() * cfl
        solver.advance(dt)

    step()  # warm-up (page faults, OpenMP team start)
    t0 = time.perf_counter()
    step()
    one = time.perf_counter() - t0
    n = max(2, min(80, int(target_seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    seconds = time.perf_counter() - t0
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            model = next(line.split(":", 1)[1].strip() for line in f if line.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(
        value=mesh.num_total_zones * n / seconds * 1e-6,
        unit="Mzps",
        cores=cores,
        kind=kind,
        sample=f"{n} RK2 steps (reference order of kernels and set_bc) of the {setup_name} workload at resolution "
               f"{resolution} ({mesh.shape[0]} x {mesh.shape[1]} zones), OpenMP on {cores} threads of {model}, {seconds:.1f} s",
    )


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from sailfish_amd.driver import timestep
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    if args.solver == "srhd":
        # BASELINE.json configs[4]: the reference's own srhd_2d problem, a relativistic shell in a homologous
        # envelope on an expanding log-spherical mesh; 4800 zones per decade = 4800 x 3273 zones on one GPU
        setup_name, setup_kwargs = "envelope-shock", dict(polar_extent=0.5)
        resolution = args.resolution or int(4800 * math.sqrt(args.gpus))
        extra = dict(fused=not args.unfused)
    elif args.solver == "dg":
        # the reference's DG-capable setup; 2048^2 zones x 18 weights on one GPU
        setup_name, setup_kwargs = "kitp-code-comparison", dict(use_dg=True)
        resolution = args.resolution or (2048 if args.strong else int(2048 * math.sqrt(args.gpus)) // 8 * 8)
        extra = dict(fused=not args.unfused)
    else:
        setup_name = "circumbinary-disk"
        resolution = args.resolution or (4096 if args.strong else default_resolution(args.gpus))
        setup_kwargs = dict(eos="isothermal" if args.solver == "iso" else "gamma-law")
        extra = dict(fused=not args.unfused)
    setup = make_setup(setup_name, **setup_kwargs)
    mesh = setup.mesh(resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=setup.start_time,
                         num_patches=world, mode="hip", **extra)
    cfl = solver.recommended_cfl

    def step():
        if args.sync_dt:
            dt = timestep(mesh, solver, cfl)  # the reference driver's literal two calls
            solver.advance(dt)
        else:
            solver.step(cfl)  # same dt and same work; the 8-byte wavespeed read-back overlaps stage 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    use_events = not args.no_events and not args.unfused
    event_log = [] if use_events else None
    stride = max(1, args.event_stride)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        # launch durations are sampled inside the timed region: the launches of every `stride`-th step
        solver.event_log = event_log if k % stride == 0 else None
        step()
    solver.event_log = event_log
    fence()
    seconds = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([seconds], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        seconds = float(t.item())

    zones = mesh.num_total_zones
    mzps = zones * args.steps / seconds * 1e-6
    BYTES_STAGE_NO_UN = 2 * 8.0 * NCONS[args.solver]
    BYTES_STAGE_UN = 3 * 8.0 * NCONS[args.solver]
    BYTES_PER_ZONE_UPDATE = BYTES_STAGE_NO_UN + BYTES_STAGE_UN
    if args.solver == "srhd":
        BYTES_PER_ZONE_UPDATE = SRHD_BYTES_PER_ZONE_UPDATE
    if args.solver == "dg":
        BYTES_PER_ZONE_UPDATE = DG_BYTES_PER_ZONE_UPDATE

    roofline = None
    kernel_ms = None
    if use_events and solver.event_log and args.solver in ("dg", "srhd"):
        dominant = "dg_advance_kernel" if args.solver == "dg" else "srhd_march_kernel"
        per_kernel = {}
        for e in solver.event_log:
            k = per_kernel.setdefault(e["kernel"], [0.0, 0, 0.0])
            k[0] += e["start"].elapsed_time(e["stop"])
            k[1] += 1
            k[2] += e["zones"] * e["bytes_per_zone"]
        ms, launches, alg_bytes = per_kernel[dominant]
        achieved = alg_bytes / (ms * 1e-3) / 1e9
        roofline = dict(
            bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4), traffic=recorded_traffic(setup.solver, resolution, world),
            kernel="sf::" + dominant, launches=launches,
            avg_launch_ms=round(ms / launches, 4), algorithmic_bytes_per_launch=round(alg_bytes / launches),
            note="fp64 VALU-bound (IEEE division and square roots kept for bit-exact results), not HBM-bound: "
                 "see DESIGN.md",
        )
        kernel_ms = {name + "_ms": round(v[0] / v[1], 4) for name, v in per_kernel.items()}
    elif use_events and solver.event_log:
        alg_bytes = 0.0
        dur_ms = 0.0
        per_kind = {False: [0.0, 0], True: [0.0, 0]}
        for e in solver.event_log:
            ms = e["start"].elapsed_time(e["stop"])
            dur_ms += ms
            alg_bytes += e["zones"] * (BYTES_STAGE_UN if e["reads_un"] else BYTES_STAGE_NO_UN)
            per_kind[e["reads_un"]][0] += ms
            per_kind[e["reads_un"]][1] += 1
        achieved = alg_bytes / (dur_ms * 1e-3) / 1e9
        traffic = recorded_traffic(setup.solver, resolution, world)
        roofline = dict(
            bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
            frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
            kernel="sf::iso_stage_kernel" if args.solver == "iso" else "sf::gam_stage_kernel",
            launches=len(solver.event_log),
            avg_launch_ms=round(dur_ms / len(solver.event_log), 4),
            algorithmic_bytes_per_launch=round(alg_bytes / len(solver.event_log)),
            note=f"launch durations from HIP events around every launch of every {stride}. timed step "
                 f"({len(solver.event_log)} launches inside the timed region)",
        )
        kernel_ms = {
            "stage_a0_ms": round(per_kind[False][0] / max(per_kind[False][1], 1), 4),
            "stage_un_ms": round(per_kind[True][0] / max(per_kind[True][1], 1), 4),
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(setup_name, setup_kwargs, resolution, args.cpu_seconds)
        except Exception as exc:  # the baseline is a report, never a reason to lose the GPU number
            cpu = dict(value=None, unit="Mzps", cores=0, kind="port", sample=f"failed: {exc!r}")

    if rank == 0:
        bytes_per_update = BYTES_PER_ZONE_UPDATE
        line = {
            "metric": f"zone-updates/sec (Mzps) fp64, {setup.solver}",
            "value": round(mzps, 2),
            "unit": "Mzps",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(seconds / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if (args.strong or args.resolution) and world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{setup.solver} {setup_name} {mesh.shape[0]}x{mesh.shape[1]} fp64, RK2, "
                            f"{world} slab patch(es), one per GPU; dt recomputed every step",
                "zones": zones,
                "zones_per_gpu": zones // world,
                "path": "reference-order kernels" if args.unfused else "fused stage kernels",
            },
            "hbm_roofline_frac_whole_step": round(mzps * 1e6 * bytes_per_update / (world * HBM_PEAK_GBS * 1e9), 4),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if kernel_ms:
            line["kernel_ms"] = kernel_ms
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
