#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "65536 3" "32768 4" "32768 3" "16384 4" "16384 3" "65536 2"; do
set -- $cfg
export CLIPMI_WIDE_SEG0=$1 CLIPMI_WIDE_SEG_RATIO=$2
echo "== seg0 $1 ratio $2: $(WC_CHECK=0 timeout -k 10 200 python3 tools/wide_check.py 10000000 256,1024 2>&1 | grep 'q/s' | tr '\n' ' ')"
done
