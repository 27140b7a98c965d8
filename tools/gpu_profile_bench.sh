#!/bin/bash
# GPU-box helper: (1) kernel-trace stats of the default bench command, (2)+(3) PMC passes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r01}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bench_stats -- python3 bench.py > gpurun_out/${tag}_bench_stats.json 2> gpurun_out/${tag}_bench_stats.err || { tail -20 gpurun_out/${tag}_bench_stats.err; exit 1; }
tail -1 gpurun_out/${tag}_bench_stats.json | cut -c1-400
f=$(find gpurun_out/${tag}_bench_stats -name "*kernel_stats*" | head -1); cut -c1-160 "$f" | head -16
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_fetch.json 2> gpurun_out/${tag}_pmc_fetch.err || { tail -20 gpurun_out/${tag}_pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_write.json 2> gpurun_out/${tag}_pmc_write.err || { tail -20 gpurun_out/${tag}_pmc_write.err; exit 1; }
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_traffic.json
# keep only the small summaries for merging back
find gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write -name "*.csv" -size +8M -delete
