#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_encode_gpu.py -q -m gpu -x > gpurun_out/r02w_test.log 2>&1; rc=$?
tail -3 gpurun_out/r02w_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --rows 1000000 --no-cpu-baseline --no-fp8 --sustained-images 0 --shard-rows 0 > gpurun_out/r02w_bench.json 2> gpurun_out/r02w_bench.err || { tail -5 gpurun_out/r02w_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r02w_bench.json').read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'], 'L14', round(d['encode_vitl14_336']['value']), d['encode_vitl14_336']['roofline']['frac'])"
