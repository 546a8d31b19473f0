#!/bin/bash
# Runs on the GPU box: the sustained RK2 pair (tools/kbench.py) over launch geometries -- waves per launch
# (SFHIP_TARGET_WAVES) and waves per workgroup (SFHIP_WAVES_PER_BLOCK).  usage: tools/sweep_waves.sh <solver>
set -u
solver=${1:-iso}
for tw in 4096 6144 8192 9216 12288 16384 24576; do
  echo "== target waves $tw"
  SFHIP_TARGET_WAVES=$tw timeout -k 10 120 python3 tools/kbench.py --solver "$solver" --reps 5 --sustained 200 2>&1 | grep "sustained RK2" || exit 1
done
for wpb in 1 2; do
  echo "== waves per block $wpb"
  SFHIP_WAVES_PER_BLOCK=$wpb timeout -k 10 120 python3 tools/kbench.py --solver "$solver" --reps 5 --sustained 200 2>&1 | grep "sustained RK2" || exit 1
done
