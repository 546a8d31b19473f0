#!/bin/bash
# Runs on the GPU box: A/B of library builds under build/ with the srhd_2d bench (envelope-shock 4800 x 3274), same box.
# usage: tools/ab_srhd.sh <lib> [<lib> ...]      (lib = file name under build/)
set -u
for lib in "$@"; do
  SFHIP_LIBRARY=$PWD/build/$lib timeout -k 10 200 python3 bench.py --solver srhd --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d.get('kernel_ms'))" || exit 1
done
