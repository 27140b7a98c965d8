#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py tests/test_kernels_gpu.py -q -m gpu -x > gpurun_out/r02l_test.log 2>&1; rc=$?
tail -4 gpurun_out/r02l_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/attn_layout_probe.py 2>&1 | tail -6
for i in 1 2; do
timeout -k 10 300 python bench.py --quick --steps 30 --rows 1000000 > gpurun_out/r02l_bench_$i.json 2> gpurun_out/r02l_bench.err || { tail -5 gpurun_out/r02l_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r02l_bench_$i.json').read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'])"
done
bash tools/gpu_stats_quick.sh r02l --rows 1000000 2>&1 | grep -E "patchify|layernorm|split|gemm256_bf16"
