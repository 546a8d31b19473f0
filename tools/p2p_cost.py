#!/usr/bin/env python3
"""
Host and device cost of the halo exchange's transport calls on the real RCCL backend, measured with ONE
process (world_size 1: RCCL accepts send/recv to the own rank inside a group), since the development box has
one GPU: `batch_isend_irecv` of two sends + two receives of one halo slab each, as a rank with two
neighbours issues per RK stage, plus the MAX all-reduce of one double.  Host time is what bounds weak
scaling once the stage kernels run concurrently with the exchange.
usage: python tools/p2p_cost.py [--nj 11584]
"""
import argparse
import os
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nj", type=int, default=11584)
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    n = 2 * (args.nj + 4) * 3
    a = [torch.zeros(n, dtype=torch.float64, device="cuda") for _ in range(4)]
    amax = torch.ones(1, dtype=torch.float64, device="cuda")
    comm = torch.cuda.Stream()

    def exchange():
        ops = [dist.P2POp(dist.isend, a[0], 0), dist.P2POp(dist.irecv, a[1], 0),
               dist.P2POp(dist.isend, a[2], 0), dist.P2POp(dist.irecv, a[3], 0)]
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            reqs = dist.batch_isend_irecv(ops)
            for r in reqs:
                r.wait()
        torch.cuda.current_stream().wait_stream(comm)

    slabs = torch.zeros(2 * n, dtype=torch.float64, device="cuda")
    gathered = torch.zeros(2 * n * dist.get_world_size(), dtype=torch.float64, device="cuda")

    def gather():  # the alternative transport: every rank's two boundary slabs in one collective
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            dist.all_gather_into_tensor(gathered, slabs)
        torch.cuda.current_stream().wait_stream(comm)

    ready, done = torch.cuda.Event(), torch.cuda.Event()

    def exchange_cached_events():  # same exchange with two events created once instead of wait_stream's per call
        ops = [dist.P2POp(dist.isend, a[0], 0), dist.P2POp(dist.irecv, a[1], 0),
               dist.P2POp(dist.isend, a[2], 0), dist.P2POp(dist.irecv, a[3], 0)]
        ready.record()
        comm.wait_event(ready)
        with torch.cuda.stream(comm):
            reqs = dist.batch_isend_irecv(ops)
            for r in reqs:
                r.wait()
            done.record()
        torch.cuda.current_stream().wait_event(done)

    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sailfish_amd.kernel.rccl import RcclComm

    direct = RcclComm.create(dist, 0, 1)
    comm_ptr = comm.cuda_stream

    def exchange_direct():  # the same exchange through RCCL called directly (kernel/rccl.py): one group, no work objects
        ready.record()
        comm.wait_event(ready)
        direct.exchange([("send", a[0], 0), ("recv", a[1], 0), ("send", a[2], 0), ("recv", a[3], 0)], comm_ptr)
        done.record(comm)
        torch.cuda.current_stream().wait_event(done)

    cur_ptr = torch.cuda.current_stream().cuda_stream
    for name, fn in (("halo exchange (2 send + 2 recv)", exchange),
                     ("same, RCCL called directly", exchange_direct),
                     ("all_reduce MAX of one double, RCCL called directly", lambda: direct.max_scalar(amax, cur_ptr)),
                     ("same, events created once", exchange_cached_events),
                     ("all_gather_into_tensor of 2 slabs", gather),
                     ("all_reduce MAX of one double", lambda: dist.all_reduce(amax, op=dist.ReduceOp.MAX))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        host = (time.perf_counter() - t0) / args.reps * 1e6
        e1.synchronize()
        print(f"{name}: host {host:.1f} us per call, device {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us per call "
              f"(back to back)", flush=True)
    # where the host time of one exchange goes
    parts = dict(ops=0.0, wait_stream=0.0, batch=0.0, wait=0.0, join=0.0)
    for _ in range(args.reps):
        t = time.perf_counter()
        ops = [dist.P2POp(dist.isend, a[0], 0), dist.P2POp(dist.irecv, a[1], 0),
               dist.P2POp(dist.isend, a[2], 0), dist.P2POp(dist.irecv, a[3], 0)]
        t1 = time.perf_counter(); parts["ops"] += t1 - t
        comm.wait_stream(torch.cuda.current_stream())
        t2 = time.perf_counter(); parts["wait_stream"] += t2 - t1
        with torch.cuda.stream(comm):
            reqs = dist.batch_isend_irecv(ops)
            t3 = time.perf_counter(); parts["batch"] += t3 - t2
            for r in reqs:
                r.wait()
            t4 = time.perf_counter(); parts["wait"] += t4 - t3
        torch.cuda.current_stream().wait_stream(comm)
        parts["join"] += time.perf_counter() - t4
    torch.cuda.synchronize()
    print("host time per exchange: " + ", ".join(f"{k} {v / args.reps * 1e6:.1f} us" for k, v in parts.items()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
