#!/bin/bash
# usage: tools_sweep.sh VAR v1 v2 ...   -- run the bench once per value of an env var, print value + key numbers
var=$1; shift
for v in "$@"; do
  export $var=$v
  out=$(timeout -k 10 120 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
  echo "$var=$v $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["kernel_ms"])')"
done
