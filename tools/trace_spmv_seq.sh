#!/bin/bash
# per-dispatch SpMV durations inside the real GMRES loop (rocprofv3 kernel trace): bash tools/trace_spmv_seq.sh <out-dir> [M]
OUT=${1:-gpurun_out/trace_spmv}; M=${2:-119}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $REPO/$OUT/t -o tr --output-format csv -- python3 $REPO/tools/probe_spmv_r2d.py $M 4 > $REPO/$OUT/run.log 2>&1 || tail -3 $REPO/$OUT/run.log
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "bcsr_spmv" in r["Kernel_Name"]]
print("spmv dispatches:", len(seq))
for s in range(0, min(len(seq), 200), 40):
    print("solve %d:" % (s // 40), " ".join("%.0f" % v for v in seq[s:s + 40]))
PY
