#!/bin/bash
# non-temporal stores in the persistent GEMM's store passes (CLIPMI_GEMM_NT_STORE, development A/B): builds the development
# library with each mask on the GPU box and times the encode step against the product library. usage: tools/gpu_gemm_nt_ab.sh "1 2 3"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { python3 tools/encode_timing.py 870 870 > gpurun_out/gnt_run.log 2>&1 || tail -3 gpurun_out/gnt_run.log; grep "^B=" gpurun_out/gnt_run.log | sed "s/^/$1 /"; }
export CLIPMI_DEV_LIB=0; run product
for m in ${1:-1 2 3}; do
  export CLIPMI_EXTRA_CXXFLAGS="-DCLIPMI_GEMM_NT_STORE=$m" CLIPMI_DEV_LIB=1
  python3 cli-p_amd/build.py --dev > gpurun_out/gnt_build_$m.log 2>&1 || { tail -5 gpurun_out/gnt_build_$m.log; exit 1; }
  run "mask=$m"
done
export CLIPMI_DEV_LIB=0; unset CLIPMI_EXTRA_CXXFLAGS; run product
