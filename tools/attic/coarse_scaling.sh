#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; export TMPDIR=/tmp
for n in 625000 1250000 2500000 5000000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs_$n -- python3 tools/topk_prof.py $n 64 51 coarse > gpurun_out/cs_$n.log 2>&1
  f=$(find gpurun_out/cs_$n -name "*kernel_stats*" | head -1)
  echo "N=$n"; grep -E "scan_coarse|rescore|select_topk|true, 2" "$f" | awk -F, '{print "   " substr($1,1,70) " calls=" $2 " avg_us=" $4/1000}'
done
