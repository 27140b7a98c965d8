#!/bin/bash
# Residual prefetch of the persistent producer (gemm256p.hpp P_TOUCH_RESID; CLIPMI_GEMM_RESID_PF = 0 off / 1 per 128-B line /
# 2 per 64 B; development library): parity of the producer with the touches on, interleaved same-box A/B of the encode
# step, and the producer's dispatch durations from rocprofv3 kernel traces. usage: tools/gpu_resid_pf_ab.sh [modes]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export CLIPMI_DEV_LIB=1
modes=${1:-0 1 2}
CLIPMI_GEMM_RESID_PF=1 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "resid" > gpurun_out/pf_tests.log 2>&1 || { tail -20 gpurun_out/pf_tests.log; exit 1; }
tail -1 gpurun_out/pf_tests.log
for r in 1 2 3; do
  for m in $modes; do
    CLIPMI_GEMM_RESID_PF=$m python3 tools/encode_ab.py 870 2>&1 | tail -1 | sed "s/^/round $r pf=$m /"
  done
done | tee gpurun_out/pf_ab.txt
for m in $modes; do
  CLIPMI_GEMM_RESID_PF=$m rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf_stats_$m -- python3 tools/encode_ab.py 870 > gpurun_out/pf_stats_$m.log 2>&1 || { tail -5 gpurun_out/pf_stats_$m.log; exit 1; }
  f=$(find gpurun_out/pf_stats_$m -name "*kernel_stats*" | head -1); cp "$f" gpurun_out/pf_kernel_stats_$m.csv
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/pf_kernel_stats_$m.csv")))
for r in rows[:8]:
    print(f"pf=$m {r['Name'][:80]:80s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f} us {float(r['Percentage']):6.2f}%")
PY
  find gpurun_out/pf_stats_$m -name "*.csv" -size +2M -delete
done | tee gpurun_out/pf_kernels.txt
