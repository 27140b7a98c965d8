#!/bin/bash
# segmented coarse scan + two batches in flight: parity first, then numbers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_topk_gpu.py -q -m gpu -x -s > gpurun_out/r02e_topk.log 2>&1; rc=$?
grep -E "re-scored|passed|failed|Error|assert" gpurun_out/r02e_topk.log | tail -12
[ $rc -ne 0 ] && { tail -30 gpurun_out/r02e_topk.log; exit $rc; }
timeout -k 10 300 python bench.py --quick --steps 30 > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err || { tail -5 gpurun_out/r02e_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02e_bench.json").read().strip().splitlines()[-1])
s=d["search"]; r=s["roofline"]
print("encode", round(d["value"]), "search", round(s["value"]), "one", s["one_batch_in_flight"], "two", s["two_batches_in_flight"])
print("scan ms", r["kernel_ms"], "GB/s", round(r["achieved"]), "surv", r["coarse_survivors_per_query"], "whole_call_frac", r["whole_call_frac"])
PY
bash tools/gpu_stats_quick.sh r02e2 2>&1 | grep -E "scan_|rescore|select|coarse_thr|fill|memset" 
