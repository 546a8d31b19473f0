"""
Two ranks sharing the one GPU of the test box (gloo transport, HIP kernels):
rehearses the multi-rank device path -- side-stream halo exchange overlapped with
the interior launch, boundary-row launches, device-side dt all-reduce -- that the
8-GPU run executes over RCCL.  (RCCL itself refuses two ranks on one device, so
the transport here is gloo; everything above the transport is identical.)
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(target, args_of_rank, world=2, timeout=100):
    """
    Start `world` ranks, collect one queue item per rank, join.  Whatever happens (a hang, a failed rank, a
    failed assertion) no child is left running: a rank that survives its test keeps the GPU and the gloo port
    and would poison every later test on the box.
    """
    import queue as queue_module

    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = [ctx.Process(target=target, args=(*args_of_rank(r), queue)) for r in range(world)]
    results = []
    try:
        for p in procs:
            p.start()
        for _ in range(world):  # drain before joining: a child blocks in exit until its queue item is read
            try:
                results.append(queue.get(timeout=timeout))
            except queue_module.Empty:
                break
        for p in procs:
            p.join(timeout=20)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert len(results) == world, f"{len(results)} of {world} ranks reported"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=5)
            if p.is_alive():
                p.kill()
                p.join(timeout=5)
    return results


def _worker(rank, world, port, name, use_step, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"  # both ranks on cuda:0
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sailfish_amd.driver import timestep
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver
        from test_golden_cpu import CASES, load

        setup_name, model, options = CASES[name]
        g = load(name)
        setup = make_setup(setup_name, **model)
        mesh = setup.mesh(int(g["resolution"]))
        s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=0.0,
                        solution=g["initial"], num_patches=world, mode="hip")
        worst = 0.0
        for k, dt_ref in enumerate(g["dt"]):
            if use_step:  # what bench.py runs: dt formed on the device behind the all-reduce
                dt = s.step(float(g["cfl"]))
            else:
                dt = timestep(mesh, s, float(g["cfl"]))
                s.advance(dt)
            worst = max(worst, abs(dt - dt_ref) / dt_ref)
        final = s.solution  # collective; gathered on rank 0 only
        if rank == 0:
            err = np.abs(final - g["final"]).max(axis=-1) / np.maximum(np.abs(g["final"]).max(axis=-1), 1e-300)
            queue.put((rank, float(err.max()), worst))
        else:
            queue.put((rank, 0.0 if final is None else float("inf"), worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_step", [False, True], ids=["advance", "step"])
@pytest.mark.parametrize("name", ["iso_disk_128_p4", "gam_disk_64", "iso_disk_rk3_48"])
def test_two_ranks_one_gpu(name, use_step):
    world = 2
    port = _free_port()
    for rank, err, dt_err in _run_ranks(_worker, lambda r: (r, world, port, name, use_step), world):
        assert err <= 1e-12, (rank, err)
        assert dt_err <= 1e-13, (rank, dt_err)


def _dg_worker(rank, world, port, name, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"  # both ranks on cuda:0
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_dg_solver import build, g_initial, zone_error

        g, mesh, s = build(name, solution=g_initial(name), num_patches=world)
        worst = 0.0
        for dt_ref in g["dt"]:
            dt = s.step(float(g["cfl"]))
            worst = max(worst, abs(dt - dt_ref) / dt_ref)
        final = s.solution  # collective; gathered on rank 0 only
        queue.put((rank, zone_error(final, g["final"]) if rank == 0 else (0.0 if final is None else float("inf")), worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["dg_disk_32", "dg_explosion_40_nolimit"])
def test_two_ranks_one_gpu_dg(name):
    """cbdisodg_2d: one-row halo slabs on the comm stream before the limiter and before the update,
    device-side dt behind the MAX all-reduce; against the reference's run."""
    world = 2
    port = _free_port()
    for rank, err, dt_err in _run_ranks(_dg_worker, lambda r: (r, world, port, name), world):
        assert err <= 1e-12, (rank, err)
        assert dt_err <= 1e-13, (rank, dt_err)


def _nccl_single_rank_worker(port, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from sailfish_amd.physics.circumbinary import Diagnostic
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver

        setup = make_setup("circumbinary-disk")
        mesh = setup.mesh(96)
        physics = dict(setup.physics, diagnostics=[Diagnostic("mass"), Diagnostic("mdot", accretion=True, which_mass=1)])
        kw = dict(setup=setup, mesh=mesh, time=0.0, num_patches=2, mode="hip")
        b = make_solver(setup.solver, physics, {}, distributed=False, **kw)
        os.environ["SFHIP_HALO_TRANSPORT"] = "torch"
        a = make_solver(setup.solver, physics, {}, distributed=True, **kw)   # every collective through torch.distributed
        del os.environ["SFHIP_HALO_TRANSPORT"]
        c = make_solver(setup.solver, physics, {}, distributed=True, **kw)   # default on RCCL: called directly, from C
        assert a.comm.enabled and a.comm.size == 1 and not b.comm.enabled and a._rccl is None and c._rccl is not None
        assert c._group is not None and a._group is None and b._group is not None
        assert c._rccl.count() == (1, 0)                                        # ncclCommCount / ncclCommUserRank
        dts_a = [a.step(0.3) for _ in range(4)]
        dts_b = [b.step(0.3) for _ in range(4)]
        dts_c = [c.step(0.3) for _ in range(4)]
        same = dts_a == dts_b == dts_c and a.maximum_wavespeed() == b.maximum_wavespeed() == c.maximum_wavespeed()
        same = same and bool(np.array_equal(c.solution, b.solution))

        # the direct transport's grouped send / receive, to the own rank (all a one-GPU box can run)
        x = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
        y = torch.zeros_like(x)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        c._rccl.exchange([("send", x[:600], 0), ("recv", y[:600], 0), ("send", x[600:], 0), ("recv", y[600:], 0)],
                         side.cuda_stream)
        side.synchronize()
        same = same and bool(torch.equal(x, y))
        # a blown-up rank must give every rank a NaN: both MAX all-reduces keep it, whatever its sign bit
        for nan in (float("nan"), -float("nan")):
            t = torch.tensor([nan], dtype=torch.float64, device="cuda")
            c._rccl.max_scalar(t, torch.cuda.current_stream().cuda_stream)
            same = same and bool(torch.isnan(t).item())
            t = torch.tensor([nan], dtype=torch.float64, device="cuda")
            same = same and bool(torch.isnan(a.comm.max_scalar(t)).item())
        t = torch.tensor([2.5], dtype=torch.float64, device="cuda")
        c._rccl.max_scalar(t, torch.cuda.current_stream().cuda_stream)
        same = same and float(a.comm.max_scalar(t).item()) == 2.5
        same = same and bool(np.array_equal(a.solution, b.solution))            # gather over the gloo side group
        same = same and a.reductions() == b.reductions()                        # per-patch sums all-gathered
        same = same and a.comm.host_group() is not None                         # RCCL is the default group, gloo beside it
        queue.put((0, 0.0 if same else float("inf"), 0.0))
    finally:
        dist.destroy_process_group()


def test_rccl_process_group_single_rank():
    """What can run of the RCCL path on one GPU: a one-rank `nccl` process group under the solver -- the MAX
    all-reduce of the device-side wavespeed slot on RCCL (through torch.distributed, and directly from the C
    step group, which is the default), NaN-keeping; the gloo side group for host gathers created beside it,
    rank-0 gather and all-gather of diagnostics -- equal to the run without torch.distributed, bit for bit.
    (Halo slabs through RCCL: test_step_group_self_exchange.)"""
    port = _free_port()
    for rank, err, _ in _run_ranks(_nccl_single_rank_worker, lambda r: (port,), world=1):
        assert err == 0.0


def _step_group_worker(kind, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import make_solver

    setup = make_setup("circumbinary-disk", eos="isothermal" if kind == "iso" else "gamma-law")
    mesh = setup.mesh(128)
    cfl = 0.3 if kind == "iso" else 0.1

    def build(num_patches, **env):
        os.environ.update(env)
        try:
            return make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0,
                               num_patches=num_patches, mode="hip", distributed=False)
        finally:
            for k in env:
                del os.environ[k]

    solvers = dict(
        one_patch=build(1),                                                    # sfhip_plan_step_rk2
        python_copies=build(4, SFHIP_NO_C_STEP="1"),                           # Python stages, device slab copies
        group_copies=build(4),                                                 # step group, device slab copies
        python_rccl=build(4, SFHIP_NO_C_STEP="1", SFHIP_SELF_EXCHANGE="1"),    # Python stages, RCCL to the own rank
        group_rccl=build(4, SFHIP_SELF_EXCHANGE="1"),                          # step group, RCCL to the own rank
        group_rccl_5=build(5, SFHIP_SELF_EXCHANGE="1"),                        # ragged slabs (26, 26, 26, 25, 25 rows)
    )
    g = solvers
    ok = g["one_patch"]._c_step and g["one_patch"]._group is None
    ok = ok and g["python_copies"]._group is None and g["python_rccl"]._group is None
    ok = ok and g["group_copies"]._group is not None and g["group_copies"]._rccl is None
    ok = ok and g["group_rccl"]._group is not None and g["group_rccl"]._rccl is not None and g["python_rccl"]._rccl is not None
    dts = {name: [s.step(cfl) for _ in range(6)] for name, s in solvers.items()}
    final = {name: s.solution for name, s in solvers.items()}
    amax = {name: s.maximum_wavespeed() for name, s in solvers.items()}
    worst = 0.0
    for name in solvers:
        if name == "group_rccl_5":
            # 128 rows in 5 slabs: dx = (xr - xl) / ni differs in the last place between patches (SURVEY section 4
            # item 5), so this one is held to the tolerance, not to the bits
            err = np.abs(final[name] - final["one_patch"]).max(axis=-1) / np.maximum(np.abs(final["one_patch"]).max(axis=-1), 1e-300)
            worst = float(err.max())
            ok = ok and worst <= 1e-12 and all(abs(a - b) <= 1e-13 * b for a, b in zip(dts[name], dts["one_patch"]))
        else:
            ok = ok and dts[name] == dts["one_patch"] and amax[name] == amax["one_patch"]
            ok = ok and bool(np.array_equal(final[name], final["one_patch"]))
    ok = ok and solvers["group_rccl"].time == solvers["one_patch"].time
    # host time of the C step with the exchange active (tools/hostprof.py prints the same for a 512 x 4096 slab)
    timing = solvers["group_rccl"]._group.step_timing()
    queue.put((0, 0.0 if ok else float("inf"), worst, timing))


@pytest.mark.parametrize("kind", ["iso", "gam"])
def test_step_group_self_exchange(kind):
    """
    The multi-rank step from C (sfhip_step_group_rk2) on what one GPU can run of it: four slab patches in one
    process whose halo slabs travel through a one-rank RCCL communicator as grouped ncclSend / ncclRecv to the own
    rank on the side stream, rows behind the halos launched there, interiors on the compute stream.  Bit for bit
    equal to the same exchange driven from Python, to device-to-device slab copies (from C and from Python) and
    to the one-patch run: same dt sequence, same final state, same reduced wavespeed.
    """
    for _, err, worst, timing in _run_ranks(_step_group_worker, lambda r: (kind,), world=1):
        assert err == 0.0, (worst, timing)


def _thread_ranks_worker(kind, world, patches_per_rank, queue, mixed=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import ctypes
    import threading
    import types

    import torch

    torch.cuda.set_device(0)
    from sailfish_amd.setups import make_setup
    from sailfish_amd.solvers import cbd_common, make_solver

    mock_path = os.path.join(HERE, "probes", "libmock_rccl.so")
    mock = ctypes.CDLL(mock_path)
    mock.mock_rccl_comm.restype = ctypes.c_void_p
    mock.mock_rccl_comm.argtypes = [ctypes.c_int, ctypes.c_int]
    mock.mock_rccl_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_long)]

    class Rendezvous:
        """torch.distributed's all_reduce for ranks that are threads: MAX / MIN of one-element tensors."""
        ReduceOp = types.SimpleNamespace(MAX="max", MIN="min")

        def __init__(self):
            self.barrier = threading.Barrier(world, timeout=60)
            self.values = [None] * world

        def get_backend(self):
            return "nccl"

        def all_reduce(self, tensor, op=None, rank=None):
            self.values[rank] = tensor.detach().cpu().clone()
            self.barrier.wait()
            stacked = torch.stack(self.values)
            result = stacked.max(dim=0).values if op == "max" else stacked.min(dim=0).values
            self.barrier.wait()
            tensor.copy_(result)

    shared = Rendezvous()
    local = threading.local()

    class RankDist:
        ReduceOp = Rendezvous.ReduceOp

        def __init__(self, rank):
            self.rank = rank

        def get_backend(self):
            return "nccl"

        def all_reduce(self, tensor, op=None):
            shared.all_reduce(tensor, op, self.rank)

    class ThreadComm(cbd_common.Comm):
        def __init__(self, rank):
            self.enabled, self.rank, self.size, self.dist = True, rank, world, RankDist(rank)
            self._host_group, self._staged = None, None

    from sailfish_amd.kernel.rccl import RcclComm

    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    mock.ncclSend.argtypes = mock.ncclRecv.argtypes = [vp, sz, ci, ci, vp, vp]
    mock.ncclAllReduce.argtypes = [vp, vp, sz, ci, ci, vp, vp]
    mock.ncclGetErrorString.argtypes, mock.ncclGetErrorString.restype = [ci], ctypes.c_char_p
    mock.sfhip_path = mock_path

    class MockRccl(RcclComm):
        """kernel/rccl.py's RcclComm (exchange, max_scalar: what the Python stages call) on the mock library."""
        def __init__(self, rank):
            super().__init__(mock, ctypes.c_void_p(mock.mock_rccl_comm(rank, world)), rank, world)

        def count(self):
            return self.size, self.rank

        def close(self):
            pass

    cbd_common.Comm.auto = classmethod(lambda cls, distributed=None: local.comm if distributed is not False else cls(False))
    cbd_common.SolverBase._create_rccl = lambda self: MockRccl(self.comm.rank)

    setup = make_setup("circumbinary-disk", eos="isothermal" if kind == "iso" else "gamma-law")
    mesh = setup.mesh(128)
    cfl = 0.3 if kind == "iso" else 0.1
    num_patches, steps = world * patches_per_rank, 6
    one = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=num_patches, mode="hip",
                      distributed=False)
    expect_dt = [one.step(cfl) for _ in range(steps)]
    expect = one.solution
    expect_amax = one.maximum_wavespeed()
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            local.comm = ThreadComm(rank)
            with torch.cuda.stream(torch.cuda.Stream()):
                s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=num_patches,
                                mode="hip")
                assert s._group is not None and s.halo_transport == "rccl" and len(s.patches) == patches_per_rank
                if mixed and rank % 2 == 1:
                    # this rank lost its step group (see SolverBase._setup_patches): it drives the stages from Python,
                    # through the same communicator, while its neighbours step from C
                    s._group = None
                dts = [s.step(cfl) for _ in range(steps)]
                amax = s.maximum_wavespeed()
                torch.cuda.synchronize()
                rows = np.concatenate([p.primitive1[2:-2, 2:-2].cpu().numpy() for p in s.patches], axis=0)
                died_at = None
                if kind == "gam":
                    # a state that blows up on ONE rank must stop ALL of them, at the same step, the way the one-rank
                    # step stops (NaN wavespeed -> NaN dt -> the Kepler solve's ValueError): the slot is reduced over
                    # the ranks as bit patterns, so the NaN reaches everybody
                    if rank == world - 1:
                        s.patches[0].primitive1[7, 9, 3] = float("nan")
                    for k in range(4):
                        try:
                            s.step(cfl)
                        except ValueError as exc:
                            assert "NaN" in str(exc) or "nan" in str(exc)
                            died_at = k
                            break
            results[rank] = (dts, amax, rows, s.time, died_at)
        except BaseException as exc:  # noqa: BLE001 -- reported by the parent
            import traceback

            errors.append((rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))))
            shared.barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=90)
    if errors or any(t.is_alive() for t in threads) or any(r is None for r in results):
        queue.put((0, float("inf"), repr(errors)[:3000], None))
        os._exit(1)  # (a rank thread may still be blocked inside the library)
    ok = all(r[0] == expect_dt and r[1] == expect_amax for r in results)
    if kind == "gam":  # every rank died, and at the same step (the step after the one that folded the NaN)
        ok = ok and all(r[4] is not None and r[4] == results[0][4] for r in results)
    else:
        ok = ok and all(r[3] == one.time for r in results)
    ok = ok and bool(np.array_equal(np.concatenate([r[2] for r in results], axis=0), expect))
    counts = (ctypes.c_long * 4)()
    mock.mock_rccl_counts(MockRccl(0).comm, counts)
    # every step: 2 stages x 2 slabs per interior seam between ranks, one all-reduce; nothing left unmatched
    seams = world - 1
    if kind == "gam":
        ok = ok and counts[0] == counts[1] and counts[3] == 0  # (the dying steps exchange too: matched, none left)
    else:
        ok = ok and counts[0] == counts[1] == steps * 2 * 2 * seams and counts[2] == steps and counts[3] == 0
    queue.put((0, 0.0 if ok else float("inf"), list(counts), None))


def _thread_ranks_worker_mixed(kind, world, patches_per_rank, queue):
    _thread_ranks_worker(kind, world, patches_per_rank, queue, mixed=True)


@pytest.mark.parametrize("kind,world,patches_per_rank,mixed", [("iso", 2, 1, False), ("iso", 3, 2, False), ("gam", 2, 2, False),
                                                              ("iso", 3, 1, True)])
def test_step_group_with_neighbours_on_other_ranks(kind, world, patches_per_rank, mixed):
    """
    sfhip_step_group_rk2 with REAL neighbours: two or three "ranks" that are threads of one process on the one GPU,
    each with its own stream, its own step group and a different rank number, exchanging their slabs and reducing
    their wavespeed through tests/probes/libmock_rccl.so -- a stand-in for the six RCCL entry points the step groups
    bind (sends and receives matched per pair of ranks in posting order, nothing moves before ncclGroupEnd, the
    all-reduce a rendezvous; RCCL itself refuses two ranks on one device).  What is under test is everything of the
    multi-rank step that is ours: which slab goes to which peer under which name, the posting order on either side,
    edges against neighbours, several patches per rank (local copies and remote slabs in one stage), streams and
    events, where the all-reduce sits, and the Python side of a rank with comm.enabled.  Equal to the one-process run
    of the same decomposition bit for bit -- dt sequence, reduced wavespeed, final state -- with exactly the expected
    number of messages and none left over.  cbdgam_2d then poisons one zone on the LAST rank: every rank's step fails
    with the Kepler solve's NaN error at the same step (the slot is reduced over the ranks as bit patterns), none
    carries on alone.  `mixed`: the odd ranks step through the Python stages instead (what a rank does whose step group
    could not be made): the same exchanges in the same order, so C-stepping and Python-stepping ranks stay in step.
    """
    args = (kind, world, patches_per_rank)
    for _, err, detail, _ in _run_ranks(_thread_ranks_worker, lambda r: args, world=1, timeout=150) if not mixed else \
            _run_ranks(_thread_ranks_worker_mixed, lambda r: args, world=1, timeout=150):
        assert err == 0.0, detail
