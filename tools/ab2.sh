#!/bin/bash
# Runs on the GPU box: A/B of library builds under build/ with tools/kbench.py --sustained (same box, same clocks).
# usage: tools/ab2.sh <solver> <lib> [<lib> ...]      (lib = file name under build/)
set -u
solver=$1; shift
for lib in "$@"; do
  echo "== $lib"
  SFHIP_LIBRARY=$PWD/build/$lib timeout -k 10 120 python3 tools/kbench.py --solver "$solver" --reps 10 --sustained 200 2>&1 | grep "sustained" || exit 1
done
