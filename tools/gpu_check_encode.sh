#!/bin/bash
# kernel + encode parity, then two short encode benches and the kernel stats (development aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
tag=${1:-chk}
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_encode_gpu.py -q -m gpu -x > gpurun_out/${tag}_test.log 2>&1; rc=$?
tail -3 gpurun_out/${tag}_test.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_stats_quick.sh ${tag} --rows 1000000 2>&1 | grep -E "attention|gemm256p|patchify"
python -c "import json; d=json.loads(open('gpurun_out/${tag}_stats.json').read().strip().splitlines()[-1]); print('bench(under rocprof)', round(d['value']), d['ms_per_step'])"
for i in 1 2; do
timeout -k 10 300 python bench.py --quick --steps 30 --rows 1000000 > gpurun_out/${tag}_bench_$i.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/${tag}_bench_$i.json').read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], round(d['roofline']['frac'],3))"
done
