#!/bin/bash
# GPU-box helper: wide pass check + timing + kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/wide_check.py 200000 65,128,200,256,300,640,1024 > gpurun_out/wide_small.log 2>&1 || { tail -30 gpurun_out/wide_small.log; exit 1; }
cat gpurun_out/wide_small.log
timeout -k 10 400 python3 tools/wide_check.py 10000000 128,256,1024 > gpurun_out/wide_10m.log 2>&1 || { tail -30 gpurun_out/wide_10m.log; exit 1; }
cat gpurun_out/wide_10m.log
CLIPMI_WIDE_WAVES=4 WC_CHECK=0 timeout -k 10 300 python3 tools/wide_check.py 10000000 256,1024 2>&1 | sed 's/^/waves4 /'
CLIPMI_WIDE=0 WC_CHECK=0 timeout -k 10 300 python3 tools/wide_check.py 10000000 1024 2>&1 | sed 's/^/old /'
cd /tmp && WC_CHECK=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_wide -o wide -- python3 $ROOT/tools/wide_check.py 10000000 1024 > $ROOT/gpurun_out/prof_wide.log 2>&1
cd $ROOT; f=$(ls gpurun_out/prof_wide/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200
