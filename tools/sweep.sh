#!/bin/bash
# Runs on the GPU box: sweeps SFHIP_ROWS_PER_CHUNK over the given values with tools/kbench.py.
# usage: tools/sweep.sh <solver> <rpc> [<rpc> ...]
set -u
solver=$1; shift
for rpc in "$@"; do
  echo "== rows_per_chunk=$rpc"
  SFHIP_ROWS_PER_CHUNK=$rpc timeout -k 10 120 python3 tools/kbench.py --solver "$solver" --reps 10 2>&1 | grep "ms" || exit 1
done
