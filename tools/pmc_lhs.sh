#!/bin/bash
# PMC pass over the Jacobian-assembly kernel (run on the GPU box): bash tools/pmc_lhs.sh <out-dir> <mode:leaf:cap>
set -e
OUT=${1:-gpurun_out/pmc_lhs}; CFG=${2:-3:16:255}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  DFL_DBG_REPS=2 rocprofv3 --kernel-trace --pmc $grp -d $REPO/$OUT/$tag -o pmc --output-format csv -- python3 $REPO/tools/dbg_patch.py 119 $CFG pmc > $REPO/$OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $REPO/$OUT/$tag.log; }
done
cd $REPO
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lhs" not in k or "pack" in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
