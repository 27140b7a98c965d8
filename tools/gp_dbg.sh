#!/bin/bash
timeout -k 10 800 python -m pytest tests/test_topk_gpu.py -x -q -m gpu -k "coarse or quantize" || exit 1
