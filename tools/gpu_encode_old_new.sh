#!/bin/bash
# Same-box A/B of the encode step: this tree against another tree of the repository copied to ab_old/ (built there:
# python ab_old/cli-p_amd/build.py), product libraries, three interleaved rounds, then rocprofv3 kernel stats of each.
# usage: tools/gpu_encode_old_new.sh [B]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-870}
export CLIPMI_DEV_LIB=0
for r in 1 2 3; do
  (cd ab_old && python3 tools/encode_ab.py $B 2>&1 | tail -1 | sed "s/^/round $r old /")
  python3 tools/encode_ab.py $B 2>&1 | tail -1 | sed "s/^/round $r new /"
done | tee gpurun_out/old_new_ab.txt
for w in old new; do
  d=$ROOT; [ $w = old ] && d=$ROOT/ab_old
  (cd $d && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/on_stats_$w -- python3 tools/encode_ab.py $B > $ROOT/gpurun_out/on_stats_$w.log 2>&1) || { tail -5 gpurun_out/on_stats_$w.log; exit 1; }
  f=$(find gpurun_out/on_stats_$w -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/on_kernel_stats_$w.csv
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/on_kernel_stats_$w.csv")))
for r in rows[:9]:
    print(f"$w {r['Name'][:80]:80s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
  find gpurun_out/on_stats_$w -name "*.csv" -size +2M -delete
done | tee gpurun_out/old_new_kernels.txt
