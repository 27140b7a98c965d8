#!/bin/bash
# development aid: parity of the GEMM kernels, then persistent-vs-plain timing and in-kernel stamps
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" || exit 1
timeout -k 10 200 python tools/gemm_persist.py 21750,3072,768,1 21750,2304,768,0 43500,3072,768,1 36928,4096,1024,1 || exit 1
timeout -k 10 200 python tools/gp_stamps.py || exit 1
