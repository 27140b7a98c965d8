#!/bin/bash
# development aid: per-K-tile cost of both 256x256 kernels (K = 768 vs 3072, same M, N)
for d in 0 2; do echo "== CLIPMI_GEMM_DBG=$d"; CLIPMI_GEMM_DBG=$d timeout -k 10 200 python tools/gemm_persist.py 21750,3072,768,1 21750,3072,3072,1 21750,3072,1536,1 || exit 1; done
