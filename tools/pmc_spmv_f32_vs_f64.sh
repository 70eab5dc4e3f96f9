#!/bin/bash
# PMC counters of the fine-level matvecs inside a PC_TWOLEVEL solve (tools/probe_twolevel_solve.py with DFL_TL_F32=1): the
# double-precision SpMV of the outer FGMRES next to the single-precision-values SpMV of the residual -- same pattern, same
# gathers, half the value bytes.  What bounds the kernel if not the bytes?   bash tools/pmc_spmv_f32_vs_f64.sh <out-dir> [M]
OUT=${1:-gpurun_out/pmc_spmv_f32}; M=${2:-119}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN2_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  DFL_TL_F32=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d $REPO/$OUT/$tag -o pmc --output-format csv -- python3 $REPO/tools/probe_twolevel_solve.py $M > $REPO/$OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -3 $REPO/$OUT/$tag.log; }
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bcsr_spmv" not in k: continue
        if int(r.get("Grid_Size", r.get("Grid_Size_X", "0")) or 0) < 10000000: continue   # fine level only (13.8M threads)
        name = "f32 values" if "f32" in k else ("f64 general form" if "<false" in k else "f64")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
