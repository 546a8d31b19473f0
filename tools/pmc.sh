#!/bin/bash
# usage: tools_pmc.sh <tag> [bench args...]   -- three separate rocprofv3 --pmc passes (SQ, FETCH, WRITE)
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/$tag/sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events "$@" > gpurun_out/$tag/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events "$@" > gpurun_out/$tag/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-events "$@" > gpurun_out/$tag/write.log 2>&1 || exit 1
echo pmc-done
