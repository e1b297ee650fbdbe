#!/bin/bash
# usage: tools/ab_trees.sh <outfile> <rounds> "<bench args>" old new ...   (GPU box)
# Same-box A/B of whole source variants: .abt/<name>/ holds the files that differ (copied over csrc/, rebuilt);
# the working tree's own sources are restored and rebuilt on exit, also when a run is cut short.
# CHAIN=1 also times the sampling-only run (tools/microbench.py chain); CHAIN_PARTS=8 adds ownership bucketing to it.
out=$1; rounds=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"
keep=$(mktemp -d /tmp/abt_keep.XXXXXX)
cp salient_plusplus_amd/csrc/*.hip salient_plusplus_amd/csrc/*.h "$keep"/
restore() { cp "$keep"/* salient_plusplus_amd/csrc/; python3 -m salient_plusplus_amd.build > /dev/null 2>&1; }
trap restore EXIT
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp "$keep"/* salient_plusplus_amd/csrc/
    [ -d ".abt/$v" ] && cp .abt/$v/* salient_plusplus_amd/csrc/
    python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build of $v failed" >> "$out"; continue; }
    timeout -k 10 400 python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('model_step') or {}
print('round $r variant $v:', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'model only', m.get('ms_per_step_model_only_resident_batch'), 'with data path', m.get('ms_per_step_with_data_path'))" >> "$out"
    if [ -n "$CHAIN" ]; then
      c=$(CHAIN_CFG=16,8 WL=S-papers timeout -k 10 300 python3 tools/microbench.py chain 2>/dev/null | grep "chain only" | sed 's/.*batches, //')
      echo "round $r variant $v: chain alone $c" >> "$out"
    fi
  done
done
cat "$out"
