#!/bin/bash
# What clock and power does the chip hold under the sustained encode step? (development aid; read-only rocm-smi queries)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
rocm-smi --showclocks --showpower --showtemp > gpurun_out/clocks_idle.txt 2>&1
timeout -k 10 120 python3 tools/encode_load.py 24 > gpurun_out/clocks_load.log 2>&1 &
pid=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp >> gpurun_out/clocks_load.txt 2>&1; sleep 1.5; done
wait $pid
tail -2 gpurun_out/clocks_load.log
grep -E "sclk|mclk|fclk|Power|Temperature" gpurun_out/clocks_idle.txt | head -12
echo ---- under load
grep -E "sclk|Power \(W\)|Average Graphics|Socket" gpurun_out/clocks_load.txt | head -24
