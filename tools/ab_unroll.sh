#!/bin/bash
# usage: tools/ab_unroll.sh <outfile> "<unrolls>" "<wg per cu list>"  (GPU box) -- delivery gather: rows in flight per lane group x workgroups per CU
trap 'python3 -m salient_plusplus_amd.build > /dev/null 2>&1' EXIT   # back to the default flag set (build.py rebuilds when the flags differ)
out=$1; mkdir -p "$(dirname "$out")"
for u in $2; do
  SPP_EXTRA_FLAGS="-DSPP_GATHER_UNROLL=$u" python3 -m salient_plusplus_amd.build --force > /dev/null 2>&1 || exit 1
  echo "== unroll $u" >> "$out"
  envs=(); for w in $3; do envs+=("SPP_GATHER_WG_PER_CU=$w"); done
  tools/ab_env.sh "$out" "--steps 192 --warmup 16" "${envs[@]}" > /dev/null
done
cat "$out"
