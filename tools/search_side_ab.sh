#!/bin/bash
# Same-box A/B of the search's side-kernel LDS budget: round-3 layout (100 KiB side kernels, 4-wave re-scoring,
# sample tied to the select's stage) against round 4's (56 KiB, 3 waves, 12 288-row sample), plus segment counts.
export CLIPMI_DEV_LIB=1
N=${1:-10000000}
run() { echo "== $*"; env "$@" python tools/search_ab2.py $N 2>&1 | grep "in flight"; }
for r in 1 2; do
run CLIPMI_SIDE_LDS_KB=100 CLIPMI_RESCORE_WAVES=4 CLIPMI_SAMPLE_ROWS=0
run CLIPMI_SIDE_LDS_KB=56 CLIPMI_RESCORE_WAVES=3
run CLIPMI_SIDE_LDS_KB=56 CLIPMI_RESCORE_WAVES=3 CLIPMI_SAMPLE_ROWS=0
run CLIPMI_SIDE_LDS_KB=56 CLIPMI_RESCORE_WAVES=3 CLIPMI_COARSE_SEGS=4
run CLIPMI_SIDE_LDS_KB=56 CLIPMI_RESCORE_WAVES=3 CLIPMI_COARSE_SEGS=5
done
