OUT=gpurun_out/r5g; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $REPO/$OUT/trace -o b -- python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-M 0 --cpu-single-M 0 --solve-to-rtol 0 --coupled-M 0 --dem-particles 0 --placement default > $REPO/$OUT/bench.json 2> $REPO/$OUT/err.txt
cd $REPO
DB=$(find $OUT/trace -name "*_results.db" | head -1)
python3 tools/rocpd_union.py $DB tet_lhs_slot -8 | cut -c1-120
cp $DB $OUT/b.db; rm -rf $OUT/trace
