#!/bin/bash
# The judged pair of one build on one box: the un-instrumented default `python bench.py` line, then the same step under
# rocprofv3 --kernel-trace --stats (side legs switched off) with the per-step kernel means.  Run on the GPU box:
#   bash tools/bench_profile.sh <out-dir>
# Copy <out-dir>/bench.json, stats/*_kernel_stats.csv, step_means.txt and bench_under_rocprof.json into profiles/.
OUT=${1:-gpurun_out/bench_profile}
REPO=$PWD
mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench.py failed"; tail -5 $OUT/bench.err; exit 1; }
python tools/show_bench.py $OUT/bench.json | head -40
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/stats -o bench --output-format csv -- python3 $REPO/bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 > $REPO/$OUT/bench_under_rocprof.json 2> $REPO/$OUT/rocprof.err || { echo "profiled bench failed"; tail -5 $REPO/$OUT/rocprof.err; exit 1; }
cd $REPO
python3 tools/trace_step_means.py $(find $OUT/stats -name "*kernel_trace.csv" | head -1) > $OUT/step_means.txt
tail -12 $OUT/step_means.txt
rm -f $(find $OUT/stats -name "*kernel_trace.csv")   # (tens of MB; the stats CSV and the step means are what is kept)
