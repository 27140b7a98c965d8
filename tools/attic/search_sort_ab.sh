#!/bin/bash
# Same-box A/B of the int8 copy's row order: rows in add order (CLIPMI_I8_SORT=0) against rows ordered by their
# largest |component|; 64-query calls, one and two in flight, then one call of 1024 queries.
for N in ${@:-10000000 12500000}; do
for r in 1 2; do
for srt in 0 1; do
  echo "== N=$N sort=$srt"; CLIPMI_I8_SORT=$srt CLIPMI_DEV_LIB=0 python tools/search_ab2.py $N 2>&1 | grep "in flight"
done; done
for srt in 0 1; do echo "== N=$N sort=$srt Q=1024"; CLIPMI_I8_SORT=$srt WC_CHECK=0 python tools/wide_check.py $N 1024 2>&1 | grep "q/s"; done
done
