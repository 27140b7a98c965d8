#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/lq
LQ_PRE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lq -- python3 tools/search_largeq.py 2000000 256 > gpurun_out/lq.log 2>&1
cat gpurun_out/lq.log | tail -4
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/lq/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "scan_coarse_kernel" in r["Kernel_Name"] and "false, true" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("main scans by (queue, stream):", collections.Counter((r["Queue_Id"], r.get("Stream_Id")) for r in rows))
for r in rows[-12:]:
    print(r["Queue_Id"], r.get("Stream_Id"), int(r["Start_Timestamp"]) % 10**9 // 1000, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 1000)
PY
find gpurun_out/lq -name "*.csv" -size +4M -delete
