#!/bin/bash
# kernel-trace stats of a short encode+search bench (development aid). usage: gpu_stats_quick.sh <tag> [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --quick --steps 30 "$@" > gpurun_out/${tag}_stats.json 2> gpurun_out/${tag}_stats.err || { tail -20 gpurun_out/${tag}_stats.err; exit 1; }
f=$(find gpurun_out/${tag}_stats -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${tag}_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:95]:95s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
find gpurun_out/${tag}_stats -name "*.csv" -size +4M -delete
