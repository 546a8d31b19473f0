"""
Multi-process (world_size 2 and 3, gloo, CPU) test of the N > 1 path: slab patches
one-or-more per rank, halo slabs as isend/irecv, interior/boundary row split, dt
as a MAX all-reduce, solution gather.  Kernels come from the CPU oracle through
tests/oracle_backend.py, so what is under test is exactly the host logic the GPU
ranks run.  Expected result: the reference's own run (golden fixture), bitwise.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, name, num_patches, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from sailfish_amd.driver import timestep
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver
        from test_golden_cpu import CASES, load

        setup_name, model, options = CASES[name]
        g = load(name)
        setup = make_setup(setup_name, **model)
        mesh = setup.mesh(int(g["resolution"]))
        s = make_solver(setup.solver, setup.physics, options, setup=setup, mesh=mesh, time=0.0,
                        num_patches=num_patches, backend=OracleBackend())
        assert len(s.patches) == num_patches // world
        ok = True
        for dt_ref, t_ref in zip(g["dt"], g["time"]):
            dt = timestep(mesh, s, float(g["cfl"]))
            ok &= dt == dt_ref
            s.advance(dt)
            ok &= s.time == t_ref
        final = s.solution  # collective; the gathered array is rank 0's, the other ranks get None
        if rank == 0:
            ok &= bool(np.array_equal(final, g["final"]))
            queue.put((rank, bool(ok), float(np.abs(final - g["final"]).max())))
        else:
            queue.put((rank, bool(ok and final is None), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,num_patches", [
    ("iso_disk_128_p4", 2, 4),   # two patches per rank: device-copy halos + cross-rank halos
    ("iso_disk_128_p4", 2, 2),   # 128 is a power of two: any slab count reproduces the fixture bitwise
    ("iso_disk_rk3_48", 3, 3),   # three ranks, RK3 (third buffer), middle rank has two neighbours
    ("gam_disk_64", 2, 2),
])
def test_slab_decomposition_over_ranks(name, world, num_patches):
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, num_patches, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)


def _srhd_worker(rank, world, port, name, num_patches, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import json

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver
        from test_srhd_solver import CASES, GOLDEN

        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        setup = make_setup(CASES[name], **json.loads(str(g["model"])))
        mesh = setup.mesh(int(g["resolution"]))
        s = make_solver(setup.solver, setup.physics, json.loads(str(g["options"])), setup=setup, mesh=mesh,
                        time=setup.start_time, num_patches=num_patches, backend=OracleBackend())
        ok = True
        for dt_ref, t_ref in zip(g["dt"], g["time"]):
            dt = mesh.min_spacing(s.time) / s.maximum_wavespeed() * float(g["cfl"])
            ok &= dt == dt_ref
            s.advance(dt)
            ok &= s.time == t_ref
        final = s.solution
        prim = s.primitive
        if num_patches == int(g["num_patches"]):
            expect, expect_prim = g["final"], g["final_primitive"]
        else:
            # the jet boundary makes patch 0 first order (srhd_2d.py:257-262), so the result depends on the
            # patch count: compare with the same decomposition run in one process
            one = make_solver(setup.solver, setup.physics, json.loads(str(g["options"])), setup=setup, mesh=mesh,
                              time=setup.start_time, num_patches=num_patches, backend=OracleBackend(),
                              distributed=False)
            for dt_ref in g["dt"]:
                one.advance(float(dt_ref))
            expect, expect_prim = one.solution, one.primitive
        if rank == 0:
            ok &= bool(np.array_equal(final, expect))
            ok &= bool(np.array_equal(prim, expect_prim))
            queue.put((rank, bool(ok), float(np.abs(final - expect).max())))
        else:
            queue.put((rank, bool(ok and final is None and prim is None), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,num_patches", [
    ("srhd_envelope_shock_p3", 3, 3),   # wavespeed MAX all-reduce, outflow at both ends
    ("srhd_envelope_jet_p2", 2, 2),     # jet boundary: wrap-around halos, both neighbours are the same peer
    ("srhd_envelope_jet_p2", 2, 4),     # ... with two patches per rank
])
def test_srhd_slab_decomposition_over_ranks(name, world, num_patches):
    """srhd_2d over ranks == the reference's run (or, for a patch count the fixture does not have, the same
    decomposition in one process), bitwise."""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_srhd_worker, args=(r, world, port, name, num_patches, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)


def _dg_worker(rank, world, port, name, num_patches, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_dg_solver import USED, build, oracle_backend

        g, mesh, s = build(name, backend=oracle_backend(), mode="host", num_patches=num_patches)
        assert len(s.patches) == num_patches // world
        initial = s.solution
        ok = bool(np.array_equal(initial, g["initial"])) if rank == 0 else initial is None
        for dt_ref, t_ref in zip(g["dt"], g["time"]):
            dt = s.step(float(g["cfl"]))
            ok &= dt == dt_ref
            ok &= s.time == t_ref
        final = s.solution
        if rank == 0:
            ok &= bool(np.array_equal(final[..., USED], g["final"][..., USED]))
            queue.put((rank, bool(ok), float(np.abs(final - g["final"])[..., USED].max())))
        else:
            queue.put((rank, bool(ok and final is None), 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world,num_patches", [
    ("dg_disk_48_p3_rk3", 3, 3),         # RK3, limiter: two one-row halo exchanges per stage, middle rank has two peers
    ("dg_disk_32", 2, 4),                # two patches per rank
    ("dg_explosion_40_nolimit", 2, 2),   # no limiter: one exchange per stage, buffer rotation keeps u^n without a copy
])
def test_dg_slab_decomposition_over_ranks(name, world, num_patches):
    """cbdisodg_2d over ranks == the reference's run, bitwise (wavespeed MAX all-reduce, device-side dt path)."""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dg_worker, args=(r, world, port, name, num_patches, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)


def _reductions_worker(rank, world, port, name, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver
        from test_golden_cpu import _load_reductions

        g, diagnostics, model = _load_reductions(name)
        setup = make_setup("circumbinary-disk", **model)
        mesh = setup.mesh(int(g["resolution"]))
        physics = dict(setup.physics, diagnostics=diagnostics)
        s = make_solver(setup.solver, physics, {}, setup=setup, mesh=mesh, time=float(g["time"]),
                        solution=g["solution"], num_patches=int(g["num_patches"]), backend=OracleBackend())
        got = np.array(s.reductions(), dtype=np.complex128)  # every rank gets every value
        queue.put((rank, bool(np.array_equal(got, g["values"])), float(np.abs(got - g["values"]).max())))
    finally:
        dist.destroy_process_group()


def test_reductions_over_ranks():
    """Solver.reductions() with one patch per rank: per-patch sums all-gathered as raw doubles, added in global
    patch order == the values the reference produced for the same state with the same patches, bitwise."""
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reductions_worker, args=(r, world, port, "iso_reductions_64_p2", queue))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)


def _nan_worker(rank, world, port, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver
        from sailfish_amd.solvers.cbd_common import Comm

        comm = Comm(True)
        ok = True
        # the reduction itself: finite values give the exact maximum; one NaN, of either sign bit, wins on every rank
        t = torch.tensor([0.25 + rank], dtype=torch.float64)
        ok &= float(comm.max_scalar(t)) == 0.25 + (world - 1)
        for nan in (float("nan"), -float("nan"), float(np.float64(0.0) * np.float64("inf"))):
            t = torch.tensor([nan if rank == 1 else 3.0 + rank], dtype=torch.float64)
            ok &= bool(torch.isnan(comm.max_scalar(t)).item())
        t = torch.tensor([float("inf") if rank == 0 else 1.0], dtype=torch.float64)
        ok &= float(comm.max_scalar(t)) == float("inf")
        # a solver whose rank 1 holds a NaN zone: every rank's maximum_wavespeed() is NaN (as numpy's max over the
        # reference's wavespeed array is), so dt is NaN everywhere and no rank carries on with a finite one
        setup = make_setup("circumbinary-disk")
        mesh = setup.mesh(32)
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=mesh, time=0.0, num_patches=world,
                        backend=OracleBackend())
        if rank == 1:
            s.patches[0].primitive1[5, 7, 2] = float("nan")
        amax = s.maximum_wavespeed()
        ok &= amax != amax
        queue.put((rank, bool(ok), 0.0))
    finally:
        dist.destroy_process_group()


def test_nan_wavespeed_reaches_every_rank():
    """The MAX all-reduce of the wavespeed keeps a NaN (reduced on the bit patterns): a blown-up rank stops all."""
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nan_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)


def _fallback_worker(rank, world, port, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle_backend import OracleBackend
        from sailfish_amd.kernel import rccl
        from sailfish_amd.setups import make_setup
        from sailfish_amd.solvers import make_solver

        setup = make_setup("circumbinary-disk")
        s = make_solver(setup.solver, setup.physics, {}, setup=setup, mesh=setup.mesh(32), time=0.0, num_patches=world,
                        backend=OracleBackend())
        closed = []

        class FakeComm:
            def close(self):
                closed.append(rank)

        def create(dist_, rank_, size_):  # rank 1 cannot join (a missing library, a failed bootstrap ...)
            if rank_ == 1:
                raise rccl.RcclError("ncclCommInitRank failed: unhandled system error (simulated)")
            return FakeComm()

        rccl.RcclComm.create = staticmethod(create)
        got = s._create_rccl()                    # collective
        ok = got is None and closed == ([rank] if rank != 1 else [])  # everyone falls back; the ranks that joined close
        rccl.RcclComm.create = staticmethod(lambda d, r, n: FakeComm())
        ok &= isinstance(s._create_rccl(), FakeComm)  # and when every rank joins, every rank keeps its communicator
        # the solver itself still steps on the default transport
        dt = s.step(0.3)
        queue.put((rank, bool(ok and dt > 0.0), 0.0))
    finally:
        dist.destroy_process_group()


def test_direct_rccl_falls_back_on_every_rank_when_one_cannot_join():
    """`SolverBase._create_rccl`: every rank tries to join the direct RCCL communicator; if ANY rank fails (here rank 1
    of 3, with the error RCCL gives for a failed bootstrap), all ranks drop it together -- the ranks that had joined
    close theirs -- and carry on through torch.distributed; a transport chosen rank by rank would deadlock the first
    exchange."""
    world = 3
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    results = [queue.get(timeout=5) for _ in range(world)]
    for p in procs:
        assert p.exitcode == 0
    for rank, ok, err in results:
        assert ok, (rank, err)
