#!/bin/bash
export CLIPMI_DEV_LIB=1   # the CLIPMI_* A/B knobs are read by the development library only (build.py --dev)
# GPU-box helper: whole encode step with the persistent GEMMs off (0) / on (1), twice, on one box
for rep in 1 2; do for d in 0 1; do echo "== CLIPMI_GEMM_PERSIST=$d"; CLIPMI_GEMM_PERSIST=$d timeout -k 10 200 python tools/encode_timing.py 435 870 || exit 1; done; done
