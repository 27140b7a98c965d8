#!/bin/bash
# ViT-L/14@336 with the persistent GEMMs off / on; B = 133 -> M = 76741 = 300 row tiles of 256 (299.8)
for d in 0 1; do echo "== CLIPMI_GEMM_PERSIST=$d"; CLIPMI_GEMM_PERSIST=$d timeout -k 10 300 python tools/encode_timing_l14.py 133 266 || exit 1; done
