#!/bin/bash
# usage: tools/prof.sh <outname> <python script> [args...]   (run on the GPU box via gpurun)
# rocprofv3 kernel trace + stats into gpurun_out/<outname>/, prints the kernel stats table.
set -e
name=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p "$ROOT/gpurun_out"
cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/$name" -- python3 "$@" > "gpurun_out/$name.log" 2>&1 || { tail -30 "gpurun_out/$name.log"; exit 1; }
f=$(find "gpurun_out/$name" -name "*kernel_stats*" | head -1)
echo "== $f"
cut -c1-220 "$f" | head -40
