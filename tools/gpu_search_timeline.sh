#!/bin/bash
# kernel timeline of the coarse search, one and two batches in flight (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-10000000}
for nfl in 1 2; do
rm -rf gpurun_out/tl$nfl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$nfl -- python3 tools/search_timeline.py run $N 64 51 $nfl > gpurun_out/tl$nfl.log 2>&1 || { tail -20 gpurun_out/tl$nfl.log; exit 1; }
grep in_flight gpurun_out/tl$nfl.log
python3 tools/search_timeline.py report gpurun_out/tl$nfl | tee gpurun_out/tl${nfl}_report.txt
find gpurun_out/tl$nfl -name "*.csv" -size +4M -delete
done
