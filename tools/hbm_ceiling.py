#!/usr/bin/env python3
"""
Measured HBM ceiling of the box ("also record a measured device-copy/triad ceiling"):
device copy b = a and triad c = a + s * b over fp64 arrays of the size of one 4096^2 cbdiso_2d state
array and of 1 GiB, timed with HIP events over 50 repetitions after 10 warm-up runs.
Prints one JSON line; numbers quoted in DESIGN.md section 5.
"""
import json

import torch


def timed(fn, reps=50, warm=10):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def main():
    out = {}
    for label, n in (("403MB", 4100 * 4100 * 3), ("1GiB", 1 << 27)):
        a = torch.rand(n, dtype=torch.float64, device="cuda")
        b = torch.rand(n, dtype=torch.float64, device="cuda")
        c = torch.empty_like(a)
        t_copy = timed(lambda: c.copy_(a))
        t_triad = timed(lambda: torch.add(a, b, alpha=1.5, out=c))
        out[label] = dict(copy_GBs=round(2 * 8 * n / t_copy / 1e9, 1), triad_GBs=round(3 * 8 * n / t_triad / 1e9, 1))
        del a, b, c
    print(json.dumps(dict(hbm_ceiling=out, device=torch.cuda.get_device_name(0))))


if __name__ == "__main__":
    main()
