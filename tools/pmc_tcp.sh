#!/bin/bash
# Runs on the GPU box: texture-path (TA / TCP) counters of the stage kernels under tools/kbench.py, one pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-pmc_tcp}
mkdir -p "$out"
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d "$out/a" -- python3 tools/kbench.py --reps 3 > "$out/a.log" 2>&1 || exit 1
echo tcp-done
