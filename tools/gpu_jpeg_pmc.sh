#!/bin/bash
# Instruction mix of the JPEG kernels (rocprofv3 --pmc, own pass, kernel trace only) on 870 files of the bench's kind and of photo-like
# content: instructions per wave-cycle say how latency-bound the Huffman kernel is. usage: bash tools/gpu_jpeg_pmc.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out
rm -rf gpurun_out/jpeg_pmc
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/jpeg_pmc -- python3 tools/jpeg_probe.py 870 > gpurun_out/jpeg_pmc.log 2>&1 || { tail -5 gpurun_out/jpeg_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, os, re
f = sorted(glob.glob("gpurun_out/jpeg_pmc/*/*counter_collection.csv"), key=os.path.getmtime)[-1]
rows = {}
for r in csv.DictReader(open(f)):
    m = re.search(r"(jpeg_\w+_kernel)", r["Kernel_Name"])
    if m:
        rows.setdefault((m.group(1), int(r["Dispatch_Id"])), {})[r["Counter_Name"]] = float(r["Counter_Value"])
by = {}
for (k, d), c in sorted(rows.items()):
    by.setdefault(k, []).append(c)
print("# rocprofv3 --pmc over tools/jpeg_probe.py 870 (launch 0: the parity batch; 1-7: noise files; 8-14: photo-like files); per launch")
for k, v in by.items():
    for label, sl in (("noise", slice(1, 8)), ("photo-like", slice(8, 15))):
        sel = v[sl]
        if not sel:
            continue
        a = {n: sum(c.get(n, 0.0) for c in sel) / len(sel) for n in sel[0]}
        wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        print(f"{k:24s} {label:10s} waves {a.get('SQ_WAVES', 0):9.0f}  SALU {a.get('SQ_INSTS_SALU', 0):12.0f}  VALU {a.get('SQ_INSTS_VALU', 0):12.0f}  LDS {a.get('SQ_INSTS_LDS', 0):11.0f}"
              f"  instructions per wave-cycle {(a.get('SQ_INSTS_SALU', 0) + a.get('SQ_INSTS_VALU', 0) + a.get('SQ_INSTS_LDS', 0)) / wc:.3f}  GUI cycles {a.get('GRBM_GUI_ACTIVE', 0):11.0f}")
PY
