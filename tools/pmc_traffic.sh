#!/bin/bash
# HBM traffic per launch of the hot kernels from PMC counters (run on the GPU box):
#   bash tools/pmc_traffic.sh <out-dir> [M]
# Two separate rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes; summary written to <out-dir>/pmc_traffic.json.
set -e
OUT=${1:-gpurun_out/pmc_traffic}; M=${2:-119}
REPO=$PWD
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $REPO/$OUT/$c -o pmc --output-format csv -- python3 $REPO/bench.py --M $M --steps 1 --warmup 0 --gmres-its 6 --cpu-M 0 --dem-particles 0 --solve-to-rtol 0 --placement default > $REPO/$OUT/$c.log 2>&1 || { echo "pass $c failed"; tail -5 $REPO/$OUT/$c.log; exit 1; }
done
cd $REPO
python3 - $OUT $M <<'PY'
import csv, glob, json, sys, collections
out, M = sys.argv[1], int(sys.argv[2])
groups = {"spmv": "bcsr_spmv_x4_kernel", "asm_lhs": "tet_lhs_slot_kernel", "asm_rhs": "tet_rhs_lane_kernel",
          "rhs_node_sum": "rhs_node_sum_kernel", "cgs_dots": "cgs_dots_stage1", "cgs_update": "cgs_update_kernel<true",
          "pc_apply": "pc_apply_kernel", "daxpy_calibration": "map3<"}
acc = {g: {"FETCH_SIZE": [], "WRITE_SIZE": []} for g in groups}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            for g, pat in groups.items():
                if pat in r["Kernel_Name"]:
                    acc[g][c].append(float(r["Counter_Value"]))
res = {"_about": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --M %d --steps 1 --warmup 0 "
                 "--gmres-its 6 --cpu-M 0; counter values are KB (bytes = value*1024); gfx950 correction per MI355X_MICROARCH.md: "
                 "FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads -> doubled ('fetch_bytes_corrected'). "
                 "Per launch = mean over the launches of that kernel in the run (cgs_*: Arnoldi steps k=0..5 only)." % M}
for g, d in acc.items():
    if not d["FETCH_SIZE"] or not d["WRITE_SIZE"]:
        continue
    fr = 1024.0 * sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
    wr = 1024.0 * sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
    res[g] = {"launches": len(d["FETCH_SIZE"]), "fetch_bytes_raw": fr, "fetch_bytes_corrected": 2.0 * fr, "write_bytes": wr,
              "hbm_bytes_per_launch": 2.0 * fr + wr}
json.dump(res, open(out + "/pmc_traffic.json", "w"), indent=1)
for g, v in res.items():
    if g != "_about":
        print("%-18s n=%3d  read %.1f MB (x2 corrected)  write %.1f MB  total %.1f MB" % (g, v["launches"], v["fetch_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
PY
