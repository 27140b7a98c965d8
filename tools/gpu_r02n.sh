#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py -q -m gpu -x > gpurun_out/r02n_test.log 2>&1; rc=$?
tail -3 gpurun_out/r02n_test.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_stats_quick.sh r02n --rows 1000000 2>&1 | grep -E "patchify|layernorm|attention|gemm256"
python -c "import json; d=json.loads(open('gpurun_out/r02n_stats.json').read().strip().splitlines()[-1]); print('bench(under rocprof)', round(d['value']), d['ms_per_step'])"
