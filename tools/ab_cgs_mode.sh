#!/bin/bash
# A/B of the CGS kernels' walk over the Krylov basis (DFL_CGS_MODE: bit 0 = the update walks the columns last to first, bit 1 =
# plain instead of nontemporal loads, bit 2 = the update's row chunks in one contiguous slab per XCD)
OUT=${1:-gpurun_out/cgs_mode}; mkdir -p $OUT
for m in ${2:-0 1 2 3}; do
  DFL_CGS_MODE=$m timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 --placement default > $OUT/mode$m.json 2> $OUT/mode$m.err || exit 1
  python - $OUT/mode$m.json $m <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["kernels"]
print("DFL_CGS_MODE=%s: %.2f ms/step, spmv %.4f ms, cgs avg %.4f ms (%.2f ms/step), drop %.17g" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], k["cgs"]["avg_ms"], k["cgs"]["total_ms_per_step"], d["gmres_residual_drop"]))
PY
done
