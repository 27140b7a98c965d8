#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
# final-build evidence: the FP8 encode step and the live-threshold search, kernel by kernel. usage: bash tools/gpu_timelines_r03d.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r03d}
rm -rf gpurun_out/etl8
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etl8 -- python3 tools/encode_timeline.py run 870 fp8 > gpurun_out/etl8.log 2>&1 || { tail gpurun_out/etl8.log; exit 1; }
python3 tools/encode_timeline.py report gpurun_out/etl8 > gpurun_out/${tag}_encode_fp8_step_timeline.txt
find gpurun_out/etl8 -name "*.csv" -size +4M -delete
rm -rf gpurun_out/etl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etl -- python3 tools/encode_timeline.py run 870 > gpurun_out/etl.log 2>&1 || { tail gpurun_out/etl.log; exit 1; }
python3 tools/encode_timeline.py report gpurun_out/etl > gpurun_out/${tag}_encode_step_timeline.txt
find gpurun_out/etl -name "*.csv" -size +4M -delete
export CLIPMI_LIVE=1
for nfl in 1 2; do
rm -rf gpurun_out/tll$nfl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tll$nfl -- python3 tools/search_timeline.py run 10000000 64 51 $nfl > gpurun_out/tll$nfl.log 2>&1 || { tail -20 gpurun_out/tll$nfl.log; exit 1; }
{ grep in_flight gpurun_out/tll$nfl.log; python3 tools/search_timeline.py report gpurun_out/tll$nfl; } > gpurun_out/${tag}_search_timeline_live_$([ $nfl = 1 ] && echo one || echo two)_in_flight.txt
find gpurun_out/tll$nfl -name "*.csv" -size +4M -delete
done
head -12 gpurun_out/${tag}_encode_fp8_step_timeline.txt; head -14 gpurun_out/${tag}_search_timeline_live_one_in_flight.txt
