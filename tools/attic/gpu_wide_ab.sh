#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
# ablations of the wide kernel (development): duration of the last segment's scan under each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for abl in 0 1 2 3; do
export CLIPMI_WIDE_ABL=$abl
echo "== ablation $abl"
rm -rf gpurun_out/pmcab
WC_CHECK=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmcab -- python3 tools/wide_check.py 10000000 1024 > gpurun_out/pmcab.log 2>&1 || { tail -5 gpurun_out/pmcab.log; exit 1; }
python3 tools/pmc_report.py gpurun_out/pmcab scan_coarse_wide 2000 | grep -E "duration|clock|busy frac|WAIT|ACTIVE_INST_ANY /"
done
