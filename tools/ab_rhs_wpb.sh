mkdir -p gpurun_out/r3n
for v in 4 8; do
  echo "== WPB=$v" >> gpurun_out/r3n/rhs_ab.txt
  DFL_RHS_WPB=$v timeout -k 10 300 python tools/rhs_wavetime.py 119 >> gpurun_out/r3n/rhs_ab.txt 2>&1 || exit 1
done
cat gpurun_out/r3n/rhs_ab.txt
