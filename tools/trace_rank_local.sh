#!/bin/bash
# Per-kernel breakdown of one rank-local step of the 8-way partition (tools/probe_rank_local.py under rocprofv3 --kernel-trace,
# digested by tools/rocpd_step.py).  Run on the GPU box:  bash tools/trace_rank_local.sh <out-dir> [M] [parts]
set -e
OUT=${1:-gpurun_out/rank_local_trace}; M=${2:-119}; PARTS=${3:-8}
REPO=$PWD
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $REPO/$OUT/trace -o rl -- python3 $REPO/tools/probe_rank_local.py $M $PARTS > $REPO/$OUT/probe_under_rocprof.txt 2>&1
cd $REPO
DB=$(find $OUT/trace -name "*_results.db" | head -1)
python3 tools/rocpd_step.py $DB > $OUT/rank_local_step_kernels.txt
cat $OUT/rank_local_step_kernels.txt
