#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"].split("(")[0]
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, counters in agg.items():
    if "stage" not in name and "wavespeed" not in name:
        continue
    print(name)
    for c, v in sorted(counters.items()):
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
