#!/bin/bash
export CLIPMI_DEV_LIB=1   # CLIPMI_WIDE2 is read by the development library only (build.py --dev)
# the two wide-pass kernels side by side (development): duration, clock, MFMA-busy and wait fractions of the scans of ONE call of
# 1 024 queries at 10 M rows. usage: tools/gpu_wide2_pmc.sh
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 0 1; do
export CLIPMI_WIDE2=$w
echo "== CLIPMI_WIDE2=$w"
rm -rf gpurun_out/pmcw2
WC_CHECK=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcw2 -- python3 tools/wide_check.py 10000000 1024 > gpurun_out/pmcw2.log 2>&1 || { tail -5 gpurun_out/pmcw2.log; exit 1; }
python3 tools/pmc_report.py gpurun_out/pmcw2 scan_coarse_wide 2000 | grep -E "duration|clock|busy frac|WAIT|ACTIVE_INST_ANY /"
python3 tools/pmc_report.py gpurun_out/pmcw2 scan_coarse_wide 0 | grep -E "duration"
done
