#!/bin/bash
# Named phases around the kernels of one transient step (run on the GPU box):  bash tools/trace_ranges.sh <out-dir> [M]
# DFL_ROCTX=1 makes the library emit roctx ranges (AssembleSystem(F) / (J), KrylovSolve, DflTimeStep, the DEM sweep);
# rocprofv3 --kernel-trace --marker-trace records them next to the kernel trace; the summary lists every range with the
# kernels that ran inside it.
set -e
OUT=${1:-gpurun_out/ranges}; M=${2:-55}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export DFL_ROCTX=1
rocprofv3 --kernel-trace --marker-trace -d $REPO/$OUT/trace -o step --output-format csv -- python3 $REPO/tools/run_transient.py $M 2 jacobi 2 40 > $REPO/$OUT/run.log 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
mk = glob.glob(out + "/trace/**/*marker_api_trace.csv", recursive=True)
kt = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
if not mk or not kt:
    print("no marker / kernel trace found under", out); sys.exit(1)
ranges = []
for r in csv.DictReader(open(mk[0])):
    name = r.get("Function") or r.get("Name") or ""
    ranges.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt[0]))]
# host-side intervals: launches are asynchronous, so a range brackets the ENQUEUE of its kernels (and, for KrylovSolve /
# DflTimeStep, the host syncs inside it), not their execution; in a trace viewer the kernels line up under the ranges by
# correlation id.  Summary: per range name, count and host time.
agg = collections.OrderedDict()
for s, e, name in sorted(ranges):
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
lines = ["%-34s x%-4d host time mean %8.3f ms" % (n, c, t / c) for n, (c, t) in agg.items()]
lines.append("kernels in the trace: %d" % len(kern))
open(out + "/ranges_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
