#!/bin/bash
# cache policy of the search's once-read streams (CLIPMI_NT_MASK): builds the development library with a mask on the
# GPU box and times it against the product library (mask = the default in csrc/topk.hip). usage: tools/gpu_nt_ab.sh "3 5 9"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() {  # label
  for q in ${NT_QS:-64 1024}; do python3 tools/search_ab2.py 10000000 $q > gpurun_out/nt_run.log 2>&1 || tail -3 gpurun_out/nt_run.log; grep "^N=" gpurun_out/nt_run.log | sed "s/^/$1 /"; done
  python3 tools/exact_timing.py 10000000 > gpurun_out/nt_run.log 2>&1 || tail -3 gpurun_out/nt_run.log; grep "^exact" gpurun_out/nt_run.log | sed "s/^/$1 /"
}
export CLIPMI_DEV_LIB=0
run "product"
for m in ${1:-3 5 9}; do
  export CLIPMI_EXTRA_CXXFLAGS="-DCLIPMI_NT_MASK=$m" CLIPMI_DEV_LIB=1
  python3 cli-p_amd/build.py --dev > gpurun_out/nt_build_$m.log 2>&1 || { tail -5 gpurun_out/nt_build_$m.log; exit 1; }
  run "mask=$m"
done
export CLIPMI_DEV_LIB=0; unset CLIPMI_EXTRA_CXXFLAGS
run "product"
