#!/bin/bash
# MFMA-busy fraction and effective clock of the persistent GEMM under the K-loop ablations (tools/gpu_gemm_kloop_abl.sh): one
# rocprofv3 --pmc pass per variant over tools/gemm_persist.py on the c_fc and c_proj shapes. usage: tools/gpu_gemm_kloop_abl_pmc.sh [variants]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export CLIPMI_DEV_LIB=1
for v in ${1:-0 1 6}; do
  if [ $v = 0 ]; then unset CLIPMI_EXTRA_CXXFLAGS; else export CLIPMI_EXTRA_CXXFLAGS="-DCLIPMI_GEMM_ABL=$v"; fi
  python3 cli-p_amd/build.py --dev > gpurun_out/kabl_build_$v.log 2>&1 || { tail -5 gpurun_out/kabl_build_$v.log; exit 1; }
  rm -rf gpurun_out/kabl_pmc_$v
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/kabl_pmc_$v -- python3 tools/gemm_persist.py 43500,3072,768,1 43500,768,3072,0 > gpurun_out/kabl_pmc_$v.log 2>&1 || { tail -5 gpurun_out/kabl_pmc_$v.log; exit 1; }
  python3 - <<PY
import csv, glob, os
d = "gpurun_out/kabl_pmc_$v"
vals = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" not in r["Kernel_Name"]: continue
        vals.setdefault((r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256p" not in r["Kernel_Name"]: continue
        dur.setdefault(r["Kernel_Name"].split("(")[0][-40:], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(set(k for k, _ in vals)):
    mf = sum(vals[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]) / len(vals[(k, "SQ_VALU_MFMA_BUSY_CYCLES")])
    gui = sum(vals[(k, "GRBM_GUI_ACTIVE")]) / len(vals[(k, "GRBM_GUI_ACTIVE")])
    us = sum(dur[k]) / len(dur[k])
    print(f"abl=$v {k}: {us:7.1f} us under the counters, MFMA-busy {mf / (1024.0 * gui / 8.0):.3f}, effective clock {gui / 8.0 / us / 1e3:.2f} GHz")
PY
  find gpurun_out/kabl_pmc_$v -name "*.csv" -size +4M -delete
done | tee gpurun_out/kloop_abl_pmc.txt
unset CLIPMI_EXTRA_CXXFLAGS; python3 cli-p_amd/build.py --dev > /dev/null 2>&1
