#!/bin/bash
# Do the slow work-space candidates of the Krylov calibration miss the address-translation caches more often?  One short
# bench run under rocprofv3 with the TCP's UTCL1 counters; every SpMV dispatch's duration (kernel trace of the same run)
# against its translation misses.   bash tools/pmc_tlb.sh <out-dir>
set -e
OUT=${1:-gpurun_out/pmc_tlb}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $REPO/$OUT/avail.txt 2>&1 || true
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_UTCL1_PERMISSION_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  DFL_WS_VERBOSE=1 DFL_WS_SETTLE_S=0 rocprofv3 --kernel-trace --pmc $grp -d $REPO/$OUT/$tag -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 1 --cpu-M 0 --coupled-M 0 --dem-particles 0 --solve-to-rtol 0 > $REPO/$OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $REPO/$OUT/$tag.log; }
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/TCP_*/")):
    cc = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "**/*kernel_trace.csv", recursive=True)
    if not cc or not kt:
        continue
    dur = {}
    for r in csv.DictReader(open(kt[0])):
        if "bcsr_spmv" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc[0])):
        if "bcsr_spmv" in r["Kernel_Name"]:
            rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    print(d)
    for k in sorted(rows, key=lambda x: int(x))[:80]:
        print("  dispatch %6s  %.4f ms  %s" % (k, dur.get(k, 0.0), "  ".join("%s %.4g" % (c, v) for c, v in sorted(rows[k].items()))))
PY
