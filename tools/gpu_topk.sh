#!/bin/bash
# GPU-box helper: topk parity tests, timing table, kernel profile
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_topk_gpu.py -x -q -m gpu > gpurun_out/topk_test.log 2>&1
rc=$?; tail -15 gpurun_out/topk_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/topk_timing.py ${1:-1000000} > gpurun_out/topk_timing.log 2>&1 || { tail -20 gpurun_out/topk_timing.log; exit 1; }
cat gpurun_out/topk_timing.log
bash tools/prof.sh prof_topk tools/topk_prof.py ${1:-1000000} 16 51
