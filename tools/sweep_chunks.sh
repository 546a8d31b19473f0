#!/bin/bash
# Sustained RK2-pair time of one patch for the library's own chunk rule ("auto") and for fixed rows per chunk.
#   tools/sweep_chunks.sh SOLVER RESOLUTION PATCHES "rows rows ..."
solver=$1; res=$2; patches=$3; shift 3
for r in auto $*; do
    if [ "$r" = auto ]; then unset SFHIP_ROWS_PER_CHUNK; else export SFHIP_ROWS_PER_CHUNK=$r; fi
    line=$(timeout -k 10 120 python3 tools/kbench.py --solver "$solver" --resolution "$res" --patches "$patches" --reps 5 --sustained 1500 2>&1 | grep "RK2 pair") || exit 1
    echo "$solver $res/$patches rows=$r $line"
done
