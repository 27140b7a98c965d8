#!/bin/bash
export CLIPMI_DEV_LIB=1   # CLIPMI_WIDE2 is read by the development library only (build.py --dev)
# kernel timeline of ONE call of 1 024 queries with either wide-pass kernel (development). usage: tools/gpu_wide2_timeline.sh [N]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-10000000}
for w in ${WIDE2_LIST:-0 1}; do
export CLIPMI_WIDE2=$w
rm -rf gpurun_out/tlw$w
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlw$w -- python3 tools/search_timeline.py run $N 1024 51 1 > gpurun_out/tlw$w.log 2>&1 || { tail -20 gpurun_out/tlw$w.log; exit 1; }
echo "== CLIPMI_WIDE2=$w"; grep in_flight gpurun_out/tlw$w.log
python3 tools/search_timeline.py report gpurun_out/tlw$w > gpurun_out/tlw${w}_report.txt; head -22 gpurun_out/tlw${w}_report.txt
find gpurun_out/tlw$w -name "*.csv" -size +4M -delete
done
