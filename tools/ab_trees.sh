#!/bin/bash
# usage: tools/ab_trees.sh <outfile> <rounds> "<bench args>" old new ...   (GPU box)
# Same-box A/B of whole source variants kept under .abt/<name>/ (files copied over csrc/, rebuilt).
out=$1; rounds=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")" /tmp/abt_keep
cp salient_plusplus_amd/csrc/*.hip salient_plusplus_amd/csrc/*.cuh salient_plusplus_amd/csrc/*.h /tmp/abt_keep/
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp .abt/$v/* salient_plusplus_amd/csrc/
    python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build of $v failed" >> "$out"; continue; }
    timeout -k 10 400 python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('model_step') or {}
print('round $r variant $v:', 'ms_per_step', round(d['ms_per_step'],4), 'model only', m.get('ms_per_step_model_only_resident_batch'), 'with data path', m.get('ms_per_step_with_data_path'))" >> "$out"
  done
done
cp /tmp/abt_keep/* salient_plusplus_amd/csrc/
cat "$out"
