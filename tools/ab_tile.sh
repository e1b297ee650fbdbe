#!/bin/bash
# usage: tools/ab_tile.sh  (GPU box): workgroup size of the two tile kernels
trap 'python3 -m salient_plusplus_amd.build > /dev/null 2>&1' EXIT   # back to the default flag set
set -e
for nt in 256 512 1024; do
  SPP_EXTRA_FLAGS="-DSPP_TILE_NT=$nt" python -m salient_plusplus_amd.build --force > /dev/null 2>&1
  echo "== TILE_NT=$nt"
  timeout -k 10 120 python tools/microbench.py chain 2>&1 | grep "chain only" | head -2
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-model-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value']/1e9)"
done
