#!/bin/bash
# Kernel breakdown of ONE PC_TWOLEVEL solve (tools/probe_twolevel_solve.py under rocprofv3 --kernel-trace; the last solve = from
# the last Galerkin re-summation to the end).  Run on the GPU box: bash tools/trace_twolevel.sh <out-dir> [M]
OUT=${1:-gpurun_out/twolevel_trace}; M=${2:-119}; REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace -d $REPO/$OUT/trace -o tl -- python3 $REPO/tools/probe_twolevel_solve.py $M > $REPO/$OUT/run.txt 2>&1
cd $REPO
grep "solve\|aggregates" $OUT/run.txt
python3 tools/rocpd_union.py $(find $OUT/trace -name "*_results.db" | head -1) galerkin_kernel -1 > $OUT/twolevel_solve_kernels.txt
cat $OUT/twolevel_solve_kernels.txt
