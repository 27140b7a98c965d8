#!/bin/bash
# Round 5: balanced tiles of the wide pass's second form (topk.hip scan_coarse_wide2_kernel). Bit-exactness at a small size,
# then the Q sweep at 10 M rows; W2MINQ lowers the smallest query count that takes the second form (default 257).
# usage: tools/gpu_wide_balance.sh [Q list] [W2MINQ list]
export CLIPMI_DEV_LIB=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
QS=${1:-65,100,128,200,256,257,300,384,512,520,640,768,777,896,1024}
timeout -k 10 300 python3 tools/wide_check.py 200000 65,128,200,256,257,300,520,640,777,1000,1024,1100 > gpurun_out/wb_small.log 2>&1 || { tail -30 gpurun_out/wb_small.log; exit 1; }
CLIPMI_WIDE2_MINQ=65 timeout -k 10 300 python3 tools/wide_check.py 200000 65,96,128,200,256 >> gpurun_out/wb_small.log 2>&1 || { tail -30 gpurun_out/wb_small.log; exit 1; }
grep -c "exact: True" gpurun_out/wb_small.log; grep -v "exact: True" gpurun_out/wb_small.log
for m in ${2:-257 65}; do
  CLIPMI_WIDE2_MINQ=$m WC_CHECK=0 timeout -k 10 500 python3 tools/wide_check.py 10000000 $QS 2>&1 | sed "s/^/minq=$m /"
done | tee gpurun_out/wb_10m.txt
