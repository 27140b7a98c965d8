#!/bin/bash
# Same-box A/B of the 10 M-row search (development aid): this build against another build of the library copied into
# the tree beforehand (AB_OLD=path/to/other/libclipmi.so, e.g. built from a stash), one and two batches in flight.
# Boxes differ by +-2.5 %, so only same-box, interleaved runs say anything about a few-percent change.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
N=${N:-10000000}
for rep in 1 2 3; do
for which in new old; do
lib=""; [ $which = old ] && lib="$ROOT/${AB_OLD:-tools/probe/libclipmi_old.so}"
[ $which = old ] && [ ! -f "$lib" ] && continue
for nfl in 1 2; do
AB_LIB=$lib AB_REPS=5 timeout -k 10 120 python3 tools/search_timeline.py run $N 64 51 $nfl 2>&1 | grep in_flight | sed "s/^/$which /" | cut -c1-110
done
done
done
