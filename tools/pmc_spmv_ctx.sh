#!/bin/bash
# PMC passes over tools/probe_spmv_r2.py (run on the GPU box): counters of the SpMV dispatches by launch context.
#   bash tools/pmc_spmv_ctx.sh <out-dir> [M]
# Each counter group is its own rocprofv3 run with --kernel-trace only (no other trace domains).
OUT=${1:-gpurun_out/pmc_spmv_ctx}; M=${2:-119}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $REPO/$OUT/counters_list.txt 2>&1
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); tag=p$i
  PROBE_EVENTS=0 PROBE_REPS=8 rocprofv3 --kernel-trace --pmc $grp -d $REPO/$OUT/$tag -o pmc --output-format csv -- python3 $REPO/tools/probe_spmv_r2.py $M > $REPO/$OUT/$tag.log 2>&1 || { echo "pass $tag ($grp) failed"; tail -3 $REPO/$OUT/$tag.log; }
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
REPS = 8
names = ["A b2b", "B after pc_apply", "C after pc_apply_scaled", "D after cgs", "E arnoldi -> new column", "F arnoldi -> fixed y"]
res = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "bcsr_spmv" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, lst in by_counter.items():
        lst.sort()
        vals = [v for _, v in lst]
        for s, nm in enumerate(names):
            seg = vals[s * REPS + 2:(s + 1) * REPS]   # drop the first two of each sequence
            if seg:
                res.setdefault(c, {})[nm] = sum(seg) / len(seg)
# durations from the kernel trace of the last pass
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/p*/**/*kernel_trace.csv", recursive=True))[-1:]:
    rows = [r for r in csv.DictReader(open(f)) if "bcsr_spmv" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows]
    for s, nm in enumerate(names):
        seg = d[s * REPS + 2:(s + 1) * REPS]
        if seg:
            res.setdefault("duration_ms(trace)", {})[nm] = sum(seg) / len(seg)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print("%-28s" % "counter" + "".join("%26s" % n for n in names))
for c, d in res.items():
    print("%-28s" % c + "".join("%26.5g" % d.get(n, float("nan")) for n in names))
PY
