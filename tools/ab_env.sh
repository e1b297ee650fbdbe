#!/bin/bash
# usage: tools/ab_env.sh <outfile> "<bench args>" "ENV=VAL ENV2=VAL" ["ENV=VAL" ...]   (GPU box)
# One un-profiled bench.py run per environment set; appends "env -> ms_per_step, deliver us" lines to <outfile>.
out=$1; args=$2; shift 2
mkdir -p "$(dirname "$out")"
for envs in "$@"; do
  ( for kv in $envs; do export "$kv"; done
    timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$envs ->', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'frac', round(d['roofline']['frac'],3))" ) >> "$out" 2>&1
done
cat "$out"
