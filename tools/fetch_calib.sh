#!/bin/bash
# FETCH_SIZE calibration per access width (run on the GPU box): bash tools/fetch_calib.sh <out-dir>
set -e
OUT=${1:-gpurun_out/fetch_calib}
REPO=$PWD
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $REPO/$OUT/pmc -o pmc --output-format csv -- /tmp/fetch_calib > $REPO/$OUT/run.log 2>&1
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]) * 1024.0)
GiB = float(1 << 30)
with open(out + "/summary.txt", "w") as fh:
    for k, v in sorted(acc.items()):
        line = "%-14s launches %d  FETCH_SIZE %.4f GiB per launch of 1 GiB streamed -> factor %.3f" % (k, len(v), sum(v) / len(v) / GiB, sum(v) / len(v) / GiB)
        print(line); fh.write(line + "\n")
PY
