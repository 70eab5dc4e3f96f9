mkdir -p gpurun_out/r4r
for i in 1 2; do
  for v in 1 0; do
    DFL_SPMV_X4=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-M 0 --cpu-single-M 0 --coupled-M 0 --dem-particles 0 --solve-to-rtol 0 > gpurun_out/r4r/b_${v}_$i.json 2> gpurun_out/r4r/b_${v}_$i.err || exit 1
    python - gpurun_out/r4r/b_${v}_$i.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d["kernels"]
print("DFL_SPMV_X4=%s: %.2f ms/step  spmv %.4f  pc_apply %.4f  cgs %.4f  b2b %.4f" % (sys.argv[2], d["ms_per_step"], k["spmv"]["avg_ms"], k["pc_apply"]["avg_ms"], k["cgs"]["avg_ms"], d["spmv_back_to_back_ms"]))
PY
  done
done
