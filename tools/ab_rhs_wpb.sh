#!/bin/bash
# A/B of the residual assembly call on the 10M-tet mesh (tools/rhs_wavetime.py): the default (8-wave workgroups claiming
# patches from LDS, LDS-transposed pack pass) against each piece switched back
OUT=${1:-gpurun_out/rhs_ab}
mkdir -p $OUT
rm -f $OUT/rhs_ab.txt
for v in "DFL_AB=default" "DFL_RHS_SUM6=0" "DFL_RHS_WPB=4" "DFL_PACK_LDS=0" "DFL_RHS_DIRECT=1"; do
  echo "== $v" >> $OUT/rhs_ab.txt
  env $v timeout -k 10 300 python tools/rhs_wavetime.py 119 >> $OUT/rhs_ab.txt 2>&1 || exit 1
done
grep "==\|Assemble" $OUT/rhs_ab.txt
