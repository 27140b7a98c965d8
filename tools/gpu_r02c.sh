#!/bin/bash
# LN-fold bring-up: kernel tests, encode parity, then an A/B of the encode step on the same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ln or split or resid" > gpurun_out/r02c_kern.log 2>&1; rc=$?
tail -15 gpurun_out/r02c_kern.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_encode_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -s > gpurun_out/r02c_enc.log 2>&1; rc=$?
grep -E "err|passed|failed|Error" gpurun_out/r02c_enc.log | tail -30
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
CLIPMI_LN_FOLD=0 timeout -k 10 300 python bench.py --quick --steps 30 > gpurun_out/r02c_bench_nofold_$i.json 2> gpurun_out/r02c_bench_nofold_$i.err || { tail -5 gpurun_out/r02c_bench_nofold_$i.err; exit 1; }
timeout -k 10 300 python bench.py --quick --steps 30 > gpurun_out/r02c_bench_fold_$i.json 2> gpurun_out/r02c_bench_fold_$i.err || { tail -5 gpurun_out/r02c_bench_fold_$i.err; exit 1; }
done
for f in gpurun_out/r02c_bench_*fold_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d['roofline']['kernel_ms'], round(d['search']['value']))"; done
