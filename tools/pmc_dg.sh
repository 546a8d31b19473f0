#!/bin/bash
# Runs on the GPU box: the SQ / FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh for library builds under build/ on the
# cbdisodg_2d bench at 2048^2, one summary line per kernel and counter.   usage: tools/pmc_dg.sh <lib> [<lib> ...]
set -u
for lib in "$@"; do
  export SFHIP_LIBRARY=$PWD/build/$lib
  tools/pmc.sh pmc_$lib --solver dg --resolution 2048 --no-traffic-pass || exit 1
  echo "== $lib"
  python3 - "gpurun_out/pmc_$lib" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, counters in agg.items():
    if "dg_advance" in name:
        print(name, {c: f"{sum(v)/len(v):.4g}" for c, v in sorted(counters.items())})
PY
done
