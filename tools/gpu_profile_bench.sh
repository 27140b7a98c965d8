#!/bin/bash
# GPU-box helper: (1) kernel-trace stats of the DEFAULT bench command, (2)+(3) HBM-traffic PMC passes,
# (4) MFMA-busy PMC pass. The program sits directly behind `--` (no env / bash -c hop), counters in their own
# runs with --kernel-trace only. Every run is the ONE-SEQUENCE form of the encode step (--encode-in-flight 1: per-kernel
# averages are then one shape, no overlap; the two-sequence headline form is timed by the un-profiled bench). usage: bash tools/gpu_profile_bench.sh <tag> [stats|pmc|mfma|all]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r02}
what=${2:-all}
if [ "$what" = all ] || [ "$what" = stats ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_bench_stats -- python3 bench.py --encode-in-flight 1 > gpurun_out/${tag}_bench_stats.json 2> gpurun_out/${tag}_bench_stats.err || { tail -20 gpurun_out/${tag}_bench_stats.err; exit 1; }
tail -1 gpurun_out/${tag}_bench_stats.json | cut -c1-400
f=$(find gpurun_out/${tag}_bench_stats -name "*kernel_stats*" | head -1); cut -c1-160 "$f" | head -24
cp "$f" gpurun_out/${tag}_bench_kernel_stats.csv
find gpurun_out/${tag}_bench_stats -name "*.csv" -size +8M -delete
fi
if [ "$what" = all ] || [ "$what" = quickstats ]; then
# the headline legs alone (no sweeps, no other batch sizes): per-name averages are then averages over ONE shape per kernel -
# the file to hold the bench line's roofline.kernel_ms against
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_quick_stats -- python3 bench.py --encode-in-flight 1 --quick --large-q 1024 > gpurun_out/${tag}_quick_under_rocprof.json 2> gpurun_out/${tag}_quick_stats.err || { tail -20 gpurun_out/${tag}_quick_stats.err; exit 1; }
tail -1 gpurun_out/${tag}_quick_under_rocprof.json | cut -c1-300
f=$(find gpurun_out/${tag}_quick_stats -name "*kernel_stats*" | head -1); cut -c1-160 "$f" | head -12
cp "$f" gpurun_out/${tag}_quick_kernel_stats.csv
find gpurun_out/${tag}_quick_stats -name "*.csv" -size +8M -delete
fi
if [ "$what" = all ] || [ "$what" = pmc ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_fetch -- python3 bench.py --encode-in-flight 1 --steps 3 --warmup 1 --quick --large-q 1024 > gpurun_out/${tag}_pmc_fetch.json 2> gpurun_out/${tag}_pmc_fetch.err || { tail -20 gpurun_out/${tag}_pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_write -- python3 bench.py --encode-in-flight 1 --steps 3 --warmup 1 --quick --large-q 1024 > gpurun_out/${tag}_pmc_write.json 2> gpurun_out/${tag}_pmc_write.err || { tail -20 gpurun_out/${tag}_pmc_write.err; exit 1; }
python3 tools/pmc_traffic.py gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write gpurun_out/${tag}_pmc_traffic.json > /dev/null || exit 1
find gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write -name "*.csv" -size +8M -delete
fi
if [ "$what" = all ] || [ "$what" = mfma ]; then
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_mfma -- python3 bench.py --encode-in-flight 1 --steps 3 --warmup 1 --quick --large-q 1024 > gpurun_out/${tag}_pmc_mfma.json 2> gpurun_out/${tag}_pmc_mfma.err || { tail -20 gpurun_out/${tag}_pmc_mfma.err; exit 1; }
python3 tools/pmc_mfma.py gpurun_out/${tag}_pmc_mfma gpurun_out/${tag}_pmc_mfma_summary.json
find gpurun_out/${tag}_pmc_mfma -name "*.csv" -size +8M -delete
fi
exit 0
