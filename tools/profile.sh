#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then four
# separate --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE for the clock under the kernel),
# all written under gpurun_out/<tag>/.  The counter passes serialise the kernels and are kept short (no clock
# ramp, 3 steps); durations quoted anywhere come from the stats pass and from bench.py itself.
# usage: tools/profile.sh <tag> [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
rm -rf "$out"   # (a tag used twice would mix the two runs' counter files)
mkdir -p "$out"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic-pass "$@" > "$out/stats.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$out/sq" -- python3 bench.py --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-events "$@" > "$out/sq.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-events "$@" > "$out/fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-events "$@" > "$out/write.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/clock" -- python3 bench.py --steps 40 --warmup 5 --ramp-seconds 0.3 --no-cpu-baseline --no-events "$@" > "$out/clock.log" 2>&1 || exit 1
grep '"metric"' "$out/stats.log" | tail -1 > "$out/bench_line.json"
echo profile-done
