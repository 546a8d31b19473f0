#!/bin/bash
# Runs on the GPU box: rows-per-chunk sweep of the iso stage kernel for one launch shape.
# usage: tools/sweep_shapes.sh <resolution> <patches> <rpc> [<rpc> ...]   (rpc 0 = the library's own choice)
set -u
res=$1; patches=$2; shift 2
for rpc in "$@"; do
  echo "== resolution=$res patches=$patches rows_per_chunk=$rpc"
  SFHIP_ROWS_PER_CHUNK=$rpc timeout -k 10 200 python3 tools/kbench.py --solver iso --resolution "$res" --patches "$patches" --reps 10 2>&1 | grep " ms" || exit 1
done
