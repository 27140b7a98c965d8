#!/bin/bash
# PMC passes over the wide search (development aid): bash tools/gpu_pmc_wide.sh [N] [Q]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-10000000}; Q=${2:-1024}
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE TCC_HIT_sum" "TCC_MISS_sum TCC_REQ_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
i=$((i+1)); rm -rf gpurun_out/pmcw$i
WC_CHECK=0 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/pmcw$i -- python3 tools/wide_check.py $N $Q > gpurun_out/pmcw$i.log 2>&1 || { tail -20 gpurun_out/pmcw$i.log; exit 1; }
python3 tools/pmc_report.py gpurun_out/pmcw$i "${KRX:-scan_coarse_wide}" ${MINUS:-2000} | tee gpurun_out/pmcw${i}_report.txt
find gpurun_out/pmcw$i -name "*.csv" -size +4M -delete
done
