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
        final = s.solution
        err = np.abs(final - g["final"]).max(axis=-1) / np.maximum(np.abs(g["final"]).max(axis=-1), 1e-300)
        queue.put((rank, float(err.max()), worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_step", [False, True], ids=["advance", "step"])
@pytest.mark.parametrize("name", ["iso_disk_128_p4", "gam_disk_64", "iso_disk_rk3_48"])
def test_two_ranks_one_gpu(name, use_step):
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, use_step, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=100)
    for p in procs:
        assert p.exitcode == 0
    for _ in range(world):
        rank, err, dt_err = queue.get(timeout=5)
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
        queue.put((rank, zone_error(s.solution, g["final"]), worst))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["dg_disk_32", "dg_explosion_40_nolimit"])
def test_two_ranks_one_gpu_dg(name):
    """cbdisodg_2d: one-row halo slabs on the comm stream before the limiter and before the update,
    device-side dt behind the MAX all-reduce; against the reference's run."""
    world = 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dg_worker, args=(r, world, port, name, queue)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=100)
    for p in procs:
        assert p.exitcode == 0
    for _ in range(world):
        rank, err, dt_err = queue.get(timeout=5)
        assert err <= 1e-12, (rank, err)
        assert dt_err <= 1e-13, (rank, dt_err)
