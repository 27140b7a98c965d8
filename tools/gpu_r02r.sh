#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_topk_gpu.py -q -m gpu -x -k "not full_size" > gpurun_out/r02r_test.log 2>&1; rc=$?
tail -3 gpurun_out/r02r_test.log
[ $rc -ne 0 ] && exit $rc
for n in 2 3 4 2 3; do
CLIPMI_BENCH_IN_FLIGHT=$n timeout -k 10 300 python bench.py --quick --steps 30 --batch 64 > gpurun_out/r02r_bench_$n.json 2> gpurun_out/r02r_bench.err || { tail -5 gpurun_out/r02r_bench.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/r02r_bench_$n.json').read().strip().splitlines()[-1]); s=d['search']; print('in flight $n:', round(s['value']), s['one_batch_in_flight'], s['two_batches_in_flight'])"
done
