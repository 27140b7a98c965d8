#!/bin/bash
# PMC evidence for DESIGN 4.4h (W fragments straight from L2): the c_fc shape on gemm256 (algo 2), its W-direct form
# (algo 5) and the persistent kernel (algo 3) under rocprofv3 counter passes (own runs, --kernel-trace only).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/wd
export TMPDIR=/tmp CLIPMI_DEV_LIB=1
rocprofv3 -L > gpurun_out/wd/counters.txt 2>&1
shape=${1:-43500x3072x768}
pass() {  # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/wd/$n -- python3 tools/attic/gemm_wd_ab.py $shape > gpurun_out/wd/$n.log 2>&1 || tail -5 gpurun_out/wd/$n.log
}
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
pass ta TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("sq", "ta", "lds", "fetch"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/wd/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "gemm256" not in k: continue
            name = "WD" if "true>" in k and "gemm256_" in k else ("persistent" if "gemm256p" in k else "gemm256")
            acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in acc.items():
        out.setdefault(n, {})[c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/wd/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find gpurun_out/wd -name "*.csv" -size +4M -delete
