#!/bin/bash
# kernel-by-kernel timelines for profiles/: 64-query search (one / two in flight), the wide search (Q = 1024), one prompt
# through the text tower, one encode step. usage: bash tools/gpu_timelines.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r03}
for nfl in 1 2; do
rm -rf gpurun_out/tl$nfl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$nfl -- python3 tools/search_timeline.py run 10000000 64 51 $nfl > gpurun_out/tl$nfl.log 2>&1 || { tail -20 gpurun_out/tl$nfl.log; exit 1; }
{ grep in_flight gpurun_out/tl$nfl.log; python3 tools/search_timeline.py report gpurun_out/tl$nfl; } > gpurun_out/${tag}_search_timeline_$([ $nfl = 1 ] && echo one || echo two)_in_flight.txt
find gpurun_out/tl$nfl -name "*.csv" -size +4M -delete
done
rm -rf gpurun_out/tlw
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlw -- python3 tools/search_timeline.py run 10000000 1024 51 1 > gpurun_out/tlw.log 2>&1 || { tail -20 gpurun_out/tlw.log; exit 1; }
{ grep in_flight gpurun_out/tlw.log; python3 tools/search_timeline.py report gpurun_out/tlw; } > gpurun_out/${tag}_search_timeline_wide_q1024.txt
find gpurun_out/tlw -name "*.csv" -size +4M -delete
rm -rf gpurun_out/tt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt -- python3 tools/text_timeline.py run > gpurun_out/tt.log 2>&1 || { tail gpurun_out/tt.log; exit 1; }
python3 tools/text_timeline.py report gpurun_out/tt all > gpurun_out/${tag}_text_one_prompt_timeline.txt
rm -rf gpurun_out/etl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/etl -- python3 tools/encode_timeline.py run 870 > gpurun_out/etl.log 2>&1 || { tail gpurun_out/etl.log; exit 1; }
python3 tools/encode_timeline.py report gpurun_out/etl > gpurun_out/${tag}_encode_step_timeline.txt
find gpurun_out/etl -name "*.csv" -size +4M -delete
tail -3 gpurun_out/${tag}_search_timeline_one_in_flight.txt; tail -2 gpurun_out/${tag}_search_timeline_wide_q1024.txt; tail -2 gpurun_out/${tag}_text_one_prompt_timeline.txt; head -4 gpurun_out/${tag}_encode_step_timeline.txt
