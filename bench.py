#!/usr/bin/env python3
"""
bench.py -- zone-updates/sec of the cbdiso_2d hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

works from a bare shell for any N: for N > 1 (and no torch.distributed environment) it starts
``python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`` as a child process --
before anything here has touched the GPU -- relays its output and exits with its code.  Launched under
torch.distributed.run directly (the round driver does that) it reads RANK / LOCAL_RANK / WORLD_SIZE from the
environment.  Rank 0 prints ONE JSON line.

A "step" is what the reference's driver does per iteration (driver.py:508-511): dt from the max-wavespeed
reduction, then one full RK2 ``solver.advance(dt)``.

Workloads
  N = 1   BASELINE.json configs[1]: circumbinary-disk 4096^2 fp64, single patch on one MI355X.
  N > 1   headline (``value``): WEAK scaling at 4096^2 zones per GPU -- a square mesh of 4096*sqrt(N) per side
          cut into N slab patches, one per rank, halo slabs and the dt MAX all-reduce on RCCL.
          Also measured in the same run and reported beside it (never as ``value``):
            "metric_config"   BASELINE.json's metric line, the 4096^2 mesh cut into N slabs (strong scaling; the
                              same object is also printed as "strong_scaling");
            "config2"         at N = 8: BASELINE.json configs[2], cbdiso_2d 16384^2 (2048 x 16384 per rank);
            "config3"         at N = 8: BASELINE.json configs[3], cbdgam_2d 8192^2 (1024 x 8192 per rank).
          "ranks" proves what ran: the number of ranks RCCL itself reports for the communicator the halos and the
          all-reduce travel on (ncclCommCount), every rank's device (PCI bus id, UUID), the transport and whether
          the one-call step group stepped the patches.
Initial data are the setup's analytic disk (no RNG, no files), resident in HBM before the clock starts.

Output fields beyond the driver's contract
  roofline       the stage kernel's algorithmic GB/s against the HBM peak; launch durations from HIP events
                 around the launches of every 4th timed step, minus the calibrated cost of an empty bracket
                 `traffic`: HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this
                 same workload, run by this command as child processes before it touches the GPU (N = 1; 3 steps
                 each); the committed profiles/traffic.json when such a pass is not possible, with the reason
                 `copy_ceiling`: what a plain device copy of one state array's bytes reaches on this box, measured by
                 this command after the timed region (N = 1), and the kernel's achieved rate over it
  roofline_valu  what actually binds: the kernel's fp64 VALU instruction stream (count per launch and the clock
                 under the kernel from two more such passes) against the issue rate of 1024 SIMDs
  cold_ms_per_step  the first K steps of a fresh solver after ONE untimed step (code-object load, exact wavespeed
                 pass), with the device's clocks still at idle: what the reference's own per-fold Mzps printout
                 (driver.py:514-517) shows for its first folds.  `ms_per_step` / `value` are the ramped figures.
  cpu_baseline   the reference's own C/OpenMP kernels (oracle/_ref, built from /root/reference by
                 oracle/build_ref.py) or, where absent, our C port (oracle/sf_oracle.c), driven in the
                 reference's order on a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

# the host driver of this pool only supports dmabuf IPC: RCCL between processes needs this (already exported
# on the boxes; kept here so that the multi-rank launch does not depend on the caller's environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 1024           # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INSTRUCTION = 4.0  # a wave64 fp64 instruction occupies its SIMD's 16 lanes for 4 cycles
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
                    help="bracket the kernel launches of every Nth timed step with HIP events")
    ap.add_argument("--strong", action="store_true",
                    help="headline = the MESH held at 4096^2 as N grows (strong scaling) instead of the zones per GPU")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="do not run the rocprofv3 --pmc child passes that measure roofline.traffic and roofline_valu (N = 1)")
    ap.add_argument("--no-extras", action="store_true", help="N > 1: measure the headline workload only")
    ap.add_argument("--halo-transport", default=None, choices=["torch", "rccl"],
                    help="N > 1: halo slabs and the wavespeed all-reduce through RCCL called directly, from C inside the "
                         "one-call step (default on an RCCL process group; sailfish_amd/kernel/rccl.py), or through "
                         "torch.distributed's own calls")
    ap.add_argument("--ramp-seconds", type=float, default=0.5,
                    help="untimed steps of the same workload run for this long BEFORE the W warmup steps, so that "
                         "the GPU's clocks have left their idle state when the (short) timed region starts; the "
                         "number of steps this took is reported as config.clock_ramp_steps")
    ap.add_argument("--base-resolution", type=int, default=4096,
                    help="side of the one-GPU mesh the scaling series are built from (4096: BASELINE.json's; smaller "
                         "values only to rehearse the N > 1 flow quickly)")
    ap.add_argument("--transport", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: all ranks share cuda:0 and exchange through gloo (rehearsal of the N > 1 flow on a "
                         "one-GPU box; RCCL refuses two ranks on one device)")
    return ap.parse_args()


def launch_ranks(args):
    """N > 1 from a bare shell: one child process per GPU through torch.distributed.run (nothing here has
    initialised the GPU: no torch import has happened in this process)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


DOMINANT_KERNEL = {"iso": "iso_stage_kernel", "gam": "gam_stage_kernel", "srhd": "srhd_march_kernel", "dg": "dg_advance_kernel"}


def counter_pass(args, resolution, counters, steps, ramp_seconds, tmp, limit=90):
    """One child run of this file under ``rocprofv3 --pmc <counters>`` (one counter block per pass).  Returns
    ({counter: [value per launch of the dominant kernel]}, [duration of those launches in ns]) or a reason string."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return "rocprofv3 not found"
    pattern = DOMINANT_KERNEL[args.solver]
    env = dict(os.environ, SFHIP_BENCH_CHILD="1", TMPDIR="/tmp")
    out = os.path.join(tmp, counters[0])
    cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.abspath(__file__), "--solver", args.solver, "--resolution", str(resolution),
           "--steps", str(steps), "--warmup", "1", "--ramp-seconds", str(ramp_seconds), "--no-cpu-baseline", "--no-events"]
    # a session of its own, so that a pass that hangs can be ended together with whatever it started
    proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                            start_new_session=True)
    try:
        status = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        proc.wait()
        return f"the {counters[0]} pass did not finish within {limit} s"
    if status != 0:
        return f"the {counters[0]} pass ended with status {status}"
    values = {c: [] for c in counters}
    spans = []
    for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if pattern in row["Kernel_Name"] and row["Counter_Name"] in values:
                    values[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Counter_Name"] == counters[0]:
                        spans.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    if not all(values.values()):
        return f"the {counters[0]} pass recorded no launch of {pattern}"
    return values, spans


def live_counters(args, resolution):
    """
    What the roofline objects need from the hardware counters, measured now: child runs of this file under
    ``rocprofv3 --pmc``, started before this process has touched the GPU -- FETCH_SIZE and WRITE_SIZE (3 steps each;
    HBM bytes per launch as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: (2 * FETCH_SIZE +
    WRITE_SIZE) * 1024), SQ_INSTS_VALU with SQ_ACTIVE_INST_VALU (3 steps: instructions per launch, cycles each) and
    GRBM_GUI_ACTIVE (40 steps after 0.3 s of load: the clock under the kernel = counter / 8 XCDs / duration, upper
    half of the launches).  Returns {"traffic": (bytes | None, note), "valu": (dict | None, note)}.
    """
    why = None
    if args.no_traffic_pass:
        why = "--no-traffic-pass"
    elif os.environ.get("SFHIP_BENCH_CHILD"):
        why = "this is a counter pass itself"
    elif "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        why = "already running under a profiler"
    if why:
        return {"traffic": (None, why), "valu": (None, why)}
    result = {}
    tmp = tempfile.mkdtemp(prefix="sfhip_pmc_", dir="/tmp")
    try:
        # (the first import of torch on a fresh box can take a minute or two: the first pass gets that long; a pass
        # that fails or hangs ends the series, so that a broken profiler costs minutes at most)
        fetch = counter_pass(args, resolution, ["FETCH_SIZE"], 3, 0, tmp, limit=240)
        if isinstance(fetch, str):
            return {"traffic": (None, fetch), "valu": (None, fetch)}
        write = counter_pass(args, resolution, ["WRITE_SIZE"], 3, 0, tmp)
        if isinstance(write, str):
            return {"traffic": (None, write), "valu": (None, write)}
        if isinstance(fetch, str) or isinstance(write, str):
            result["traffic"] = (None, fetch if isinstance(fetch, str) else write)
        else:
            f, w = fetch[0]["FETCH_SIZE"], write[0]["WRITE_SIZE"]
            result["traffic"] = (round((2.0 * sum(f) / len(f) + sum(w) / len(w)) * 1024.0),
                                 "measured by this command: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes of "
                                 f"the same workload (3 steps, {len(f)} launches each), (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                                 "per launch")
        sq = counter_pass(args, resolution, ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"], 3, 0, tmp)
        clock = counter_pass(args, resolution, ["GRBM_GUI_ACTIVE"], 40, 0.3, tmp) if not isinstance(sq, str) else sq
        if isinstance(sq, str) or isinstance(clock, str):
            result["valu"] = (None, sq if isinstance(sq, str) else clock)
        else:
            insts, active = sq[0]["SQ_INSTS_VALU"], sq[0]["SQ_ACTIVE_INST_VALU"]
            ghz = sorted(v / 8.0 / ns for v, ns in zip(clock[0]["GRBM_GUI_ACTIVE"], clock[1]) if ns > 0)
            ghz = ghz[len(ghz) // 2:]  # the first launches of the pass run on the clock ramp
            result["valu"] = (dict(insts_per_launch=round(sum(insts) / len(insts)),
                                   cycles_per_inst=round(4.0 * sum(active) / sum(insts), 3),
                                   clock_ghz=round(sum(ghz) / len(ghz), 3)),
                              "measured by this command: rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU (3 steps) and "
                              f"--pmc GRBM_GUI_ACTIVE (40 steps, upper half of {len(clock[1])} launches) child passes")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return result


def recorded(table, solver, resolution, world):
    """A per-launch figure of the dominant kernel from the committed PMC passes (tools/profile.sh ->
    tools/collect_profile.py -> profiles/<table>.json), or None for a configuration that was not profiled."""
    path = os.path.join(ROOT, "profiles", table + ".json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f).get(f"{solver}_{resolution}_{world}")


def default_resolution(gpus, base=4096):
    n = base * math.sqrt(gpus)
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


class Ranks:
    """The process's place in the job and the two collectives the bench itself needs."""

    def __init__(self, args):
        import torch

        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.transport = args.transport
        if self.transport == "gloo":
            os.environ["LOCAL_RANK"] = "0"  # every rank on cuda:0 (the solvers read LOCAL_RANK too)
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist

            self.dist = dist
            if self.transport == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64,
                              device="cpu" if self.transport == "gloo" else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def bracket_overhead_ms(torch, samples=40):
    """What a pair of HIP events around NOTHING measures on the launch stream: the part of every bracket that
    is not the kernel (the command processor's event handling)."""
    torch.cuda.synchronize()
    pairs = []
    for _ in range(samples):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    times = sorted(a.elapsed_time(b) for a, b in pairs)
    return times[len(times) // 2]


def transport_proof(ranks, solver):
    """What carried the halos and the dt all-reduce of this solver, as the libraries themselves report it."""
    torch = ranks.torch
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    mine = dict(rank=ranks.rank, device=torch.cuda.current_device(),
                pci="%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                        getattr(props, "pci_device_id", 0)),
                uuid=str(getattr(props, "uuid", "")), name=props.name)
    rccl = getattr(solver, "_rccl", None)
    if rccl is not None:
        mine["rccl_comm_count"], mine["rccl_comm_rank"] = rccl.count()  # ncclCommCount / ncclCommUserRank
    mine["halo_transport"] = getattr(solver, "halo_transport", None)
    mine["step_group"] = getattr(solver, "_group", None) is not None
    if ranks.dist is None:
        return [mine]
    everyone = [None] * ranks.world
    ranks.dist.all_gather_object(everyone, mine)
    return everyone


def copy_ceiling_gbs(torch, nbytes, reps=30, warm=5):
    """What a plain device copy of an array of `nbytes` reaches on this box right now (read + write bytes per second):
    the practical ceiling of any kernel that streams HBM, beside the 8 TB/s peak the roofline fraction is quoted on."""
    n = max(1, int(nbytes) // 8)
    a = torch.rand(n, dtype=torch.float64, device="cuda")
    b = torch.empty_like(a)
    for _ in range(warm):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    e1.synchronize()
    seconds = e0.elapsed_time(e1) * 1e-3 / reps
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * 8.0 * n / seconds / 1e9


def measure(args, ranks, solver_kind, resolution, steps, warmup, use_events, measure_cold=False):
    """Build the workload, run `warmup` untimed and `steps` timed steps between fences; MAX seconds over ranks."""
    torch = ranks.torch
    from sailfish_amd.driver import timestep
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    if solver_kind == "srhd":
        # BASELINE.json configs[4]: the reference's own srhd_2d problem, a relativistic shell in a homologous
        # envelope on an expanding log-spherical mesh; 4800 zones per decade = 4800 x 3273 zones on one GPU
        setup_name, setup_kwargs = "envelope-shock", dict(polar_extent=0.5)
    elif solver_kind == "dg":
        # the reference's DG-capable setup; 2048^2 zones x 18 weights on one GPU
        setup_name, setup_kwargs = "kitp-code-comparison", dict(use_dg=True)
    else:
        setup_name = "circumbinary-disk"
        setup_kwargs = dict(eos="isothermal" if solver_kind == "iso" else "gamma-law")
    setup = make_setup(setup_name, **setup_kwargs)
    mesh = setup.mesh(resolution)
    solver = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=setup.start_time,
                         num_patches=ranks.world, mode="hip", fused=not args.unfused)
    cfl = solver.recommended_cfl

    def step():
        if args.sync_dt:
            dt = timestep(mesh, solver, cfl)  # the reference driver's literal two calls
            solver.advance(dt)
        else:
            solver.step(cfl)  # same dt and same work; no host round trip on the critical path

    proof = transport_proof(ranks, solver)
    # Cold figure: one untimed step (lazy code-object load, the exact wavespeed pass of a new solver), then the
    # same K steps between the same fences, before any clock ramp.
    cold_ms = None
    if measure_cold:
        step()
        ranks.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ranks.fence()
        cold_ms = ranks.max_over_ranks(time.perf_counter() - t0) / steps * 1e3
    ramp_steps = 0
    if args.ramp_seconds > 0.0:
        # The device leaves its idle clocks only after some tens of milliseconds of load: with the round driver's
        # 5 warmup + 20 timed steps (14 ms of work) the timed region would otherwise sit on the ramp itself
        # (0.69 against 0.57 ms per step, measured).  Untimed steps of the same workload for about ramp_seconds
        # come first; every rank takes the same number of them (there are collectives inside a step).
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        per_step = max((time.perf_counter() - t0) / 5, 1e-5)
        ramp_steps = int(ranks.max_over_ranks(float(min(5000, int(args.ramp_seconds / per_step)))))
        if ranks.rank == 0 and os.environ.get("SFHIP_BENCH_VERBOSE"):
            print(f"[bench] clock ramp: {ramp_steps} steps after 5 steps at {per_step * 1e3:.3f} ms", file=sys.stderr, flush=True)
        for _ in range(ramp_steps):
            step()
        torch.cuda.synchronize()
        ramp_steps += 5
    for _ in range(warmup):
        step()
    if ranks.rank == 0 and os.environ.get("SFHIP_BENCH_VERBOSE"):
        print(f"[bench] ramp and warmup done at {resolution}", file=sys.stderr, flush=True)
    event_log = [] if use_events else None
    redos_before = getattr(solver, "wavespeed_redos", 0)
    # every 4th step by default, but no more than 8 sampled steps in all: a bracketed step goes through the
    # Python stages instead of the one-call step, which a long run of small steps would otherwise feel
    stride = max(1, args.event_stride, -(-steps // 8))
    ranks.fence()
    t0 = time.perf_counter()
    for k in range(steps):
        # launch durations are sampled inside the timed region: the launches of every `stride`-th step
        solver.event_log = event_log if k % stride == 0 else None
        step()
    ranks.fence()
    seconds = ranks.max_over_ranks(time.perf_counter() - t0)
    solver.event_log = None
    result = dict(resolution=resolution, setup=setup, setup_name=setup_name, setup_kwargs=setup_kwargs, mesh=mesh, seconds=seconds,
                  steps=steps, zones=mesh.num_total_zones, mzps=mesh.num_total_zones * steps / seconds * 1e-6,
                  ms_per_step=seconds / steps * 1e3, event_log=event_log, stride=stride, ramp_steps=ramp_steps,
                  patch_shape=tuple(solver.patches[0].shape), redos=getattr(solver, "wavespeed_redos", 0) - redos_before,
                  cold_ms_per_step=cold_ms, proof=proof)
    if use_events and event_log:
        result["bracket_overhead_ms"] = bracket_overhead_ms(torch)
    del solver
    torch.cuda.empty_cache()
    return result


def roofline_of(args, run, world):
    """roofline / roofline_valu / kernel_ms objects from the event brackets of a measured run (or Nones)."""
    log = run["event_log"]
    if not log:
        return None, None, None
    solver = run["setup"].solver
    resolution = run["resolution"]
    over = run["bracket_overhead_ms"]
    ncons = NCONS[args.solver]
    bytes_no_un, bytes_un = 2 * 8.0 * ncons, 3 * 8.0 * ncons
    per_kernel = {}
    for e in log:
        ms = max(e["start"].elapsed_time(e["stop"]) - over, 0.0)
        if "kernel" in e:  # srhd / dg logs name their kernels and bytes
            name, alg = e["kernel"], e["zones"] * e["bytes_per_zone"]
        else:
            name = "stage_un" if e["reads_un"] else "stage_a0"
            alg = e["zones"] * (bytes_un if e["reads_un"] else bytes_no_un)
        k = per_kernel.setdefault(name, [0.0, 0, 0.0])
        k[0] += ms
        k[1] += 1
        k[2] += alg
    if args.solver in ("dg", "srhd"):
        dominant = "dg_advance_kernel" if args.solver == "dg" else "srhd_march_kernel"
        ms, launches, alg_bytes = per_kernel[dominant]
        kernel = "sf::" + dominant
    else:
        ms = sum(v[0] for v in per_kernel.values())
        launches = sum(v[1] for v in per_kernel.values())
        alg_bytes = sum(v[2] for v in per_kernel.values())
        kernel = "sf::iso_stage_kernel" if args.solver == "iso" else "sf::gam_stage_kernel"
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    traffic, traffic_source = run.get("live_traffic", (None, None))
    if traffic is None:
        why = traffic_source
        traffic = recorded("traffic", solver, resolution, world)
        traffic_source = None if traffic is None else (
            "profiles/traffic.json (committed PMC passes of this shape)" + (f"; no live pass: {why}" if why else ""))
    roofline = dict(
        bound="hbm", achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
        frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
        kernel=kernel, launches=launches, avg_launch_ms=round(ms / launches, 4),
        algorithmic_bytes_per_launch=round(alg_bytes / launches),
        binding={"iso": "balanced between the fp64 instruction stream (0.41 ms per RK2 pair alone) and the memory traffic "
                        "(0.45 ms alone, at 0.85-0.9 of the copy ceiling); together 0.50: DESIGN.md sections 3.1 and 5",
                 "gam": "its fp64 instruction stream at two waves per SIMD (dependent chains, not issue slots alone: "
                        "see roofline_valu and DESIGN.md sections 3.1 and 5); its memory traffic alone takes 0.58 ms"}.get(
                     args.solver, "fp64 VALU issue, not HBM: see roofline_valu and DESIGN.md"),
        note=f"launch durations: HIP events around every launch of every {run['stride']}. timed step "
             f"({len(log)} launches inside the timed region), minus the median of 40 empty brackets "
             f"({over * 1e3:.1f} us) measured on the same stream right after the timed region; a bracketed launch "
             "runs alone (the events keep it from overlapping its neighbours' ramp-up and drain), so the two "
             "stages' durations can add up to slightly more than ms_per_step",
    )
    roofline["traffic_note"] = traffic_source or (f"no PMC pass was recorded for {solver} at {resolution}^2 on {world} "
                                                  "GPU(s): profiles/traffic.json holds the profiled shapes")
    valu = None
    insts, valu_source = run.get("live_valu", (None, None))
    if insts is None:
        why = valu_source
        insts = recorded("valu", solver, resolution, world)
        valu_source = "profiles/valu.json (committed PMC passes of this shape)" + (f"; no live pass: {why}" if why else "")
    if insts:
        clock_ghz = insts.get("clock_ghz", 2.1)
        count = insts["insts_per_launch"]  # SQ_INSTS_VALU per launch, averaged over the step's launches
        rate = count / (ms / launches * 1e-3)
        peak = SIMDS * clock_ghz * 1e9 / VALU_CYCLES_PER_INSTRUCTION
        valu = dict(
            bound="valu", achieved=round(rate / 1e9, 1), peak=round(peak / 1e9, 1), unit="G wave-instructions/s",
            frac=round(rate / peak, 4), insts_per_launch=count, cycles_per_instruction=insts.get("cycles_per_inst"),
            clock_ghz=clock_ghz,
            note="SQ_INSTS_VALU per launch / this run's launch duration, against 1024 SIMDs x clock / 4 cycles per "
                 "wave64 fp64 instruction; the clock is the one measured under this kernel (GRBM_GUI_ACTIVE / 8 / "
                 "duration)",
            counters=valu_source,
        )
    kernel_ms = {name + "_ms": round(v[0] / v[1], 4) for name, v in per_kernel.items()}
    return roofline, valu, kernel_ms


def summary(run, world, bytes_per_update):
    proof = run.get("proof") or []
    return dict(
        halo_transport=proof[0].get("halo_transport") if proof else None,
        step_group=all(r.get("step_group") for r in proof) if proof else None,
        workload=f"{run['setup'].solver} {run['setup_name']} {run['mesh'].shape[0]}x{run['mesh'].shape[1]} fp64, RK2, "
                 f"{world} slab patch(es) of {run['patch_shape'][0]}x{run['patch_shape'][1]}, one per GPU",
        value=round(run["mzps"], 2), unit="Mzps", ms_per_step=round(run["ms_per_step"], 4), steps=run["steps"],
        hbm_roofline_frac_whole_step=round(run["mzps"] * 1e6 * bytes_per_update / (world * HBM_PEAK_GBS * 1e9), 4),
    )


def ranks_object(proof, world, transport):
    """The N ranks as the libraries report them: distinct devices, and RCCL's own count of the communicator."""
    counts = sorted({r.get("rccl_comm_count") for r in proof if r.get("rccl_comm_count") is not None})
    return dict(
        world_size=world,
        rccl_comm_count=counts[0] if len(counts) == 1 else (counts or None),  # ncclCommCount on every rank
        rccl_comm_ranks=sorted(r["rccl_comm_rank"] for r in proof if "rccl_comm_rank" in r) or None,
        distinct_devices=len({(r["pci"], r["uuid"]) for r in proof}),
        devices=[dict(rank=r["rank"], pci=r["pci"], uuid=r["uuid"], name=r["name"]) for r in proof],
        halo_transport=sorted({str(r.get("halo_transport")) for r in proof}),
        step_group=all(r.get("step_group") for r in proof),
        process_group="gloo (rehearsal: every rank on cuda:0)" if transport == "gloo" else ("RCCL" if world > 1 else None),
    )


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    if os.environ.get("SFHIP_BENCH_WATCHDOG"):
        # a stuck run says where: every thread's Python stack to stderr after this many seconds, then exit
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["SFHIP_BENCH_WATCHDOG"]), exit=True)
    if args.halo_transport:
        os.environ["SFHIP_HALO_TRANSPORT"] = args.halo_transport
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    if args.solver == "srhd":
        resolution = args.resolution or int(4800 * math.sqrt(world))
    elif args.solver == "dg":
        resolution = args.resolution or (2048 if args.strong else int(2048 * math.sqrt(world)) // 8 * 8)
    else:
        base = args.base_resolution
        resolution = args.resolution or (base if args.strong else default_resolution(world, base))
    strong = (args.strong or args.resolution is not None) and world > 1

    use_events = not args.no_events and not args.unfused
    # the counter passes are child processes: they come first, while this process has not touched the GPU
    if world == 1 and use_events:
        counters_now = live_counters(args, resolution)
    else:
        counters_now = {"traffic": (None, "N > 1" if world > 1 else None), "valu": (None, "N > 1" if world > 1 else None)}
    ranks = Ranks(args)
    head = measure(args, ranks, args.solver, resolution, args.steps, args.warmup, use_events,
                   measure_cold=not os.environ.get("SFHIP_BENCH_CHILD"))
    head["live_traffic"] = counters_now["traffic"]
    head["live_valu"] = counters_now["valu"]
    bytes_per_update = 5 * 8.0 * NCONS[args.solver]
    if args.solver == "srhd":
        bytes_per_update = SRHD_BYTES_PER_ZONE_UPDATE
    if args.solver == "dg":
        bytes_per_update = DG_BYTES_PER_ZONE_UPDATE
    roofline, valu, kernel_ms = roofline_of(args, head, world)

    # N > 1: the other scaling series, measured in the same job (never `value`)
    extras = {}
    if world > 1 and args.solver == "iso" and not args.no_extras and args.resolution is None:
        plans = {}
        if not args.strong:
            plans["strong_scaling"] = base          # BASELINE.json's metric: the 4096^2 mesh over N GPUs
        else:
            plans["weak_scaling"] = default_resolution(world, base)
        if world == 8:
            plans["config2"] = 4 * base             # BASELINE.json configs[2]: 16384^2, 2048 x 16384 per rank
            plans["config3"] = 2 * base             # BASELINE.json configs[3]: cbdgam_2d 8192^2, 1024 x 8192 per rank
        for name, n in plans.items():
            kind = "gam" if name == "config3" else "iso"
            try:
                run = measure(args, ranks, kind, n, args.steps, min(args.warmup, 5), False)
                extras[name] = summary(run, world, 5 * 8.0 * NCONS[kind])
            except Exception as exc:  # an extra is a report, never a reason to lose the headline
                extras[name] = dict(failed=repr(exc))
        if "strong_scaling" in extras:
            # BASELINE.json's metric is quoted on this series: the same object under the name that says so
            extras["metric_config"] = extras["strong_scaling"]

    if roofline is not None and world == 1 and not os.environ.get("SFHIP_BENCH_CHILD"):
        # the same device, right after the timed region: a copy of one state array's worth of bytes
        try:
            ceiling = copy_ceiling_gbs(ranks.torch, head["zones"] * 8.0 * NCONS[args.solver])
            roofline["copy_ceiling"] = dict(
                value=round(ceiling, 1), unit="GB/s", frac_of_peak=round(ceiling / HBM_PEAK_GBS, 4),
                achieved_over_copy=round(roofline["achieved"] / ceiling, 4),
                note="torch device copy of one state array's bytes, read + write, 30 repetitions, measured by this "
                     "command after the timed region: what streaming reaches on this box; `frac` stays the fraction "
                     "of the 8 TB/s peak")
        except Exception as exc:  # a report, never a reason to lose the line
            roofline["copy_ceiling"] = dict(failed=repr(exc))

    cpu = None
    if ranks.rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(head["setup_name"], head["setup_kwargs"], resolution, args.cpu_seconds)
        except Exception as exc:  # the baseline is a report, never a reason to lose the GPU number
            cpu = dict(value=None, unit="Mzps", cores=0, kind="port", sample=f"failed: {exc!r}")

    if ranks.rank == 0:
        setup, mesh = head["setup"], head["mesh"]
        line = {
            "metric": f"zone-updates/sec (Mzps) fp64, {setup.solver}" + ("" if world == 1 else (
                "; value = the 4096^2 mesh over N GPUs (BASELINE.json's metric line)" if strong and args.resolution is None else
                "; value = a fixed mesh over N GPUs" if strong else
                "; value = weak scaling at 4096^2 zones per GPU, BASELINE.json's 4096^2-over-N-GPUs line is metric_config")),
            "value": round(head["mzps"], 2),
            "unit": "Mzps",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{setup.solver} {head['setup_name']} {mesh.shape[0]}x{mesh.shape[1]} fp64, RK2, "
                            f"{world} slab patch(es) of {head['patch_shape'][0]}x{head['patch_shape'][1]}, one per "
                            f"GPU; dt recomputed every step"
                            + ("" if world == 1 else
                               ("; headline = strong scaling of the 4096^2 mesh" if strong else
                                "; headline = weak scaling at 4096^2 zones per GPU (strong scaling and config 2 "
                                "are reported beside it)")),
                "zones": head["zones"],
                "zones_per_gpu": head["zones"] // world,
                "path": "reference-order kernels" if args.unfused else "fused stage kernels",
                "clock_ramp_steps": head["ramp_steps"],
                "repeated_stage0": head["redos"],  # timed steps redone because the wavespeed fell below its filter
                "transport": None if world == 1 else (
                    ("RCCL called directly" + (" from C (step group)" if all(r.get("step_group") for r in head["proof"]) else "")
                     if head["proof"][0].get("halo_transport") == "rccl" else "RCCL through torch.distributed")
                    if args.transport == "nccl" else "gloo, all ranks on cuda:0"),
            },
            "hbm_roofline_frac_whole_step": round(head["mzps"] * 1e6 * bytes_per_update / (world * HBM_PEAK_GBS * 1e9), 4),
            "roofline": roofline,
            "roofline_valu": valu,
            "cpu_baseline": cpu,
            "cold_ms_per_step": None if head["cold_ms_per_step"] is None else round(head["cold_ms_per_step"], 4),
            "cold_value": None if head["cold_ms_per_step"] is None else round(head["zones"] / head["cold_ms_per_step"] * 1e-3, 2),
            "ranks": ranks_object(head["proof"], world, args.transport),
        }
        if kernel_ms:
            line["kernel_ms"] = kernel_ms
        line.update(extras)
        print(json.dumps(line), flush=True)

    ranks.close()


if __name__ == "__main__":
    main()
