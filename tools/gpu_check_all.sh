#!/bin/bash
# kernel + encode + topk parity, then the default bench (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
tag=${1:-chk}
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_encode_gpu.py tests/test_topk_gpu.py -q -m gpu -x > gpurun_out/${tag}_test.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_test.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_full.sh ${tag} bench
