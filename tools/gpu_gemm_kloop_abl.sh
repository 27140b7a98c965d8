#!/bin/bash
# K-loop ablations of the persistent GEMM (gemm256p.hpp CLIPMI_GEMM_ABL, timing only): builds the development library with each
# variant ON THE GPU BOX and times the four ViT-B/32 shapes stand-alone (bias-only / QuickGELU store pass, tools/gemm_persist.py);
# the default library first and last. What it answers: is the K-loop bound by LDS fragment reads (a 128 x 128 wave
# tile would cut them by a third: variant 1) or by the operand traffic from beyond the CU (variants 2, 6)?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
export CLIPMI_DEV_LIB=1
SHAPES="43500,2304,768,0 43500,768,768,0 43500,3072,768,1 43500,768,3072,0"
run() { python3 tools/gemm_persist.py $SHAPES 2>&1 | grep "^M=" | sed "s/^/abl=$1 /"; }
for v in 0 ${1:-1 2 6 7} 0; do
  if [ $v = 0 ]; then unset CLIPMI_EXTRA_CXXFLAGS; else export CLIPMI_EXTRA_CXXFLAGS="-DCLIPMI_GEMM_ABL=$v"; fi
  python3 cli-p_amd/build.py --dev > gpurun_out/kabl_build_$v.log 2>&1 || { tail -5 gpurun_out/kabl_build_$v.log; exit 1; }
  run $v
done | tee gpurun_out/kloop_abl.txt
