#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
# GPU-box helper: wide pass check + timing + kernel timeline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python3 tools/wide_check.py 200000 1,16,64,65,128,200,256,300,640,1024 > gpurun_out/wide_small.log 2>&1 || { tail -30 gpurun_out/wide_small.log; exit 1; }
cat gpurun_out/wide_small.log
timeout -k 10 400 python3 tools/wide_check.py 10000000 64,128,256,1024 > gpurun_out/wide_10m.log 2>&1 || { tail -30 gpurun_out/wide_10m.log; exit 1; }
cat gpurun_out/wide_10m.log
[ -n "$WIDE_AB" ] && { CLIPMI_WIDE_WAVES=4 WC_CHECK=0 timeout -k 10 300 python3 tools/wide_check.py 10000000 256,1024 2>&1 | sed 's/^/waves4 /'; }
for Q in ${TLQ:-1024}; do rm -rf gpurun_out/tlw$Q; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlw$Q -- python3 tools/search_timeline.py run 10000000 $Q 51 1 > gpurun_out/tlw$Q.log 2>&1 || { tail -20 gpurun_out/tlw$Q.log; exit 1; }; grep in_flight gpurun_out/tlw$Q.log; python3 tools/search_timeline.py report gpurun_out/tlw$Q > gpurun_out/tlw${Q}_report.txt; head -20 gpurun_out/tlw${Q}_report.txt; find gpurun_out/tlw$Q -name "*.csv" -size +4M -delete; done
