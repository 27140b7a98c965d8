#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/l14_stats -- python3 tools/encode_timing_l14.py 266 > gpurun_out/l14_stats.log 2> gpurun_out/l14_stats.err || { tail -20 gpurun_out/l14_stats.err; exit 1; }
tail -2 gpurun_out/l14_stats.log
f=$(find gpurun_out/l14_stats -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/l14_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/l14_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
find gpurun_out/l14_stats -name "*.csv" -size +4M -delete
