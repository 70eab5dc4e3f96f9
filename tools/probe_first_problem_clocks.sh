export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
mkdir -p gpurun_out/r4i
( for i in $(seq 1 60); do echo "t=$(date +%s.%N)" ; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|socclk" | head -3; sleep 0.5; done ) > gpurun_out/r4i/clocks.txt 2>&1 &
SM=$!
timeout -k 10 300 python tools/probe_rank_local.py 119 1 > gpurun_out/r4i/probe.txt 2>&1
kill $SM 2>/dev/null
grep "per step" gpurun_out/r4i/probe.txt | cut -c100-220
grep -i "sclk" gpurun_out/r4i/clocks.txt | awk '{print $NF, $(NF-1)}' | tr '\n' ' ' | cut -c1-1500
