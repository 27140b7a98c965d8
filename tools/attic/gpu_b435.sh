#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
for v in 2 1; do
CLIPMI_GEMM_PERSIST=$v timeout -k 10 300 python bench.py --quick --steps 40 --rows 1000000 --batch 435 > gpurun_out/b435_$v.json 2> gpurun_out/b435.err || { tail -5 gpurun_out/b435.err; exit 1; }
python -c "import json; d=json.loads(open('gpurun_out/b435_$v.json').read().strip().splitlines()[-1]); print('B=435 persist_mode $v', round(d['value']), d['ms_per_step'])"
done
done
timeout -k 10 300 python -m pytest tests/test_encode_gpu.py -q -m gpu -x -k "batch_invariance or matches_oracle" 2>&1 | tail -2
