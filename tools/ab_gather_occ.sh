#!/bin/bash
# usage: tools/ab_gather_occ.sh <outfile> "<unrolls>" "<wg per cu list>"  (GPU box): gather alone, rows in flight per lane group x workgroups per CU
trap 'python3 -m salient_plusplus_amd.build > /dev/null 2>&1' EXIT   # back to the default flag set (build.py rebuilds when the flags differ)
out=$1; mkdir -p "$(dirname "$out")"
for u in $2; do
  SPP_EXTRA_FLAGS="-DSPP_GATHER_UNROLL=$u" python3 -m salient_plusplus_amd.build --force > /dev/null 2>&1 || exit 1
  echo "== unroll $u" >> "$out"
  for w in $3; do SPP_GATHER_WG_PER_CU=$w timeout -k 10 120 python3 tools/gather_occupancy.py 2>/dev/null >> "$out"; done
done
cat "$out"
