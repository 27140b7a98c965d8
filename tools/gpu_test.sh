#!/bin/bash
# GPU-box helper: run the given pytest targets with -m gpu, log to gpurun_out/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
name=$1; shift
timeout -k 10 900 python -m pytest "$@" -q -m gpu -s > gpurun_out/$name.log 2>&1
rc=$?
grep -E "passed|failed|error|err |Error|assert" gpurun_out/$name.log | tail -40
exit $rc
