#!/usr/bin/env python3
"""
Gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (development tool):
    python tools/kernel_gaps.py <dir with *_kernel_trace.csv> [last N launches, default 400]
prints, per kernel name, the mean duration and the mean idle time between its start and the end of the launch
before it -- what the device waits for the host (or for the command processor) between dependent launches.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    paths = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    rows = []
    for path in paths:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
    rows.sort()
    rows = rows[-last:]
    dur, gap = defaultdict(list), defaultdict(list)
    for (s0, e0, _), (s1, e1, name) in zip(rows, rows[1:]):
        dur[name].append(e1 - s1)
        gap[name].append(s1 - e0)
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f"{len(rows)} launches over {span * 1e-3:.1f} us, device busy {100.0 * busy / span:.1f} %")
    for name in dur:
        d, g = dur[name], gap[name]
        print(f"  {name:60s} n {len(d):5d}  duration {sum(d) / len(d) * 1e-3:8.2f} us  idle before {sum(g) / len(g) * 1e-3:7.2f} us")


if __name__ == "__main__":
    main()
