#!/bin/bash
# Runs on the GPU box: one --pmc pass with GRBM_GUI_ACTIVE (effective clock = value / 8 / kernel wall time,
# MI355X_MICROARCH.md "DVFS give-back") next to the VALU counters, and a kernel trace for the wall times.
# usage: tools/pmc_clock.sh <tag> [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$out/clock" -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-events "$@" > "$out/clock.log" 2>&1 || exit 1
echo clock-done
