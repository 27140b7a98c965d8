#!/bin/bash
# same-box A/B of the search over scan LDS paddings all:main:side (KiB), two batches in flight
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out
N=${N:-10000000}
for rep in 1 2 3; do
for cfg in ${CFGS:-0:90:100 0:96:100 0:104:100 0:112:100 0:128:100 0:159:100}; do
IFS=: read pad padm side <<< "$cfg"
for nfl in ${NFL:-2}; do
CLIPMI_SCAN_PAD_KB=$pad CLIPMI_MAIN_PAD_KB=$padm CLIPMI_SIDE_KB=$side AB_REPS=5 timeout -k 10 120 python3 tools/search_timeline.py run $N 64 51 $nfl 2>&1 | grep in_flight | sed "s/^/all=$pad main=$padm side=$side /" | cut -c1-120
done
done
done
