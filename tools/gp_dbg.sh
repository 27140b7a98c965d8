#!/bin/bash
timeout -k 10 200 python tools/gemm_persist.py 43500,768,768,2 43500,768,3072,2 87000,768,768,2 87000,768,3072,2 || exit 1
