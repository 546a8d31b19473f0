"""Can two RCCL ranks share ONE GPU on this box?  (tools/probe: a question put to the machine, not product code.)

Run as:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
             tools/probe/rccl_two_ranks_one_gpu.py
Both ranks bind cuda:0.  NCCL refuses this ("Duplicate GPU detected"); the probe records what RCCL does here.
"""
import os
import sys
import torch
import torch.distributed as dist


def main():
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        x = torch.full((4,), float(rank + 1), device="cuda", dtype=torch.float64)
        dist.all_reduce(x, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_reduce MAX -> {x.tolist()}", flush=True)
        y = torch.full((8,), float(rank), device="cuda", dtype=torch.float64)
        z = torch.empty_like(y)
        peer = 1 - rank
        ops = [dist.P2POp(dist.isend, y, peer), dist.P2POp(dist.irecv, z, peer)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        print(f"rank {rank}: received {z.tolist()}", flush=True)
    except Exception as e:  # what the probe is for
        print(f"rank {rank}: {type(e).__name__}: {str(e)[:600]}", flush=True)
        sys.exit(3)
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
