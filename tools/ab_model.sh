#!/bin/bash
# usage: tools/ab_model.sh <outfile> <reps> "<bench args>" "ENV=VAL ..." ["ENV=VAL ..." ...]   (GPU box)
# Interleaved repetitions of the model-step leg of bench.py: model only / with the data path feeding it / epoch time.
# BENCH_EXTRA="--prime 16" inside an environment set adds bench arguments for that set only (quote it without spaces: BENCH_EXTRA=--prime=16).
out=$1; reps=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"
for r in $(seq 1 $reps); do
  k=0
  for envs in "$@"; do
    k=$((k+1))
    ( for kv in $envs; do export "$kv"; done
      timeout -k 10 400 python3 bench.py $args $BENCH_EXTRA --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['model_step']
print('cfg$k rep$r [$envs]', 'data_ms', round(d['ms_per_step'],4), 'model_only', round(m['ms_per_step_model_only_resident_batch'],4), 'with_data', round(m['ms_per_step_with_data_path'],4), 'epoch_s', round(d['epoch_time_s_with_model_step'],4))" ) >> "$out" 2>&1
  done
done
cat "$out"
