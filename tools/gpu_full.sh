#!/bin/bash
# full GPU suite, then the default bench command (what the driver runs)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
tag=${1:-r02}
if [ "$2" != bench ]; then
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_gputest.log 2>&1; rc=$?
tail -4 gpurun_out/${tag}_gputest.log
[ $rc -ne 0 ] && exit $rc
fi
t0=$(date +%s)
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -20 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench wall seconds: $(( $(date +%s) - t0 ))"
python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms", round(d["ms_per_step"],3), "frac", round(d["roofline"]["frac"],3), "whole", round(d["roofline"]["whole_step_frac"],3), "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
for k in ("encode_sustained","encode_fp8","encode_vitl14_336"):
    print(k, {kk: (round(v,3) if isinstance(v,float) else v) for kk,v in d[k].items() if kk in ("value","seconds","ms_per_step","images_per_gpu","min_cosine_to_bf16_path","whole_step_frac")})
print("l14 roofline", d["encode_vitl14_336"]["roofline"]["frac"])
for k in ("search","search_shard_12p5m"):
    s=d[k]; print(k, round(s["value"]), round(s["ms_per_step"],3), s["batches_in_flight"], "scan_ms", round(s["roofline"]["kernel_ms"],3), "frac", round(s["roofline"]["frac"],3), "whole", round(s["roofline"]["whole_call_frac"],3), "surv", s["roofline"]["coarse_survivors_per_query"])
print("cpu", {k:(round(v,2) if isinstance(v,float) else v) for k,v in d["cpu_baseline"].items() if k!="sample"})
print("cpu search", d["search"]["cpu_baseline"]["value"])
PY
