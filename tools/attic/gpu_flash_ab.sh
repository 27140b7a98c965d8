#!/bin/bash
# flash attention (long sequences): parity tests, then the ViT-L/14@336 step time and per-kernel stats (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_encode_gpu.py -q -m gpu -x -k "attention or l14 or b16 or toyl14 or vitl14" > gpurun_out/flash_test.log 2>&1; rc=$?
tail -3 gpurun_out/flash_test.log
[ $rc -ne 0 ] && exit $rc
for v in 1 2; do
timeout -k 10 200 python tools/encode_timing_l14.py 266 2>&1 | tail -1
done
bash tools/gpu_l14_stats.sh 2>&1 | grep -E "flash|ViT-L"
