#!/bin/bash
# development aid: whole encode step with the persistent GEMM off (0) / loader-only DMA (1) / shared DMA (2), twice
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm256p" || exit 1
for rep in 1 2; do for d in 0 1 2; do echo "== CLIPMI_GEMM_PERSIST=$d"; CLIPMI_GEMM_PERSIST=$d timeout -k 10 200 python tools/encode_timing.py 435 870 || exit 1; done; done
