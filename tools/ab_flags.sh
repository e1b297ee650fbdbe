#!/bin/bash
# usage: tools/ab_flags.sh <outfile> "<bench args>" "<extra hipcc flags>" ...   (GPU box; "-" = the default flag set)
# One rebuild + bench.py run per flag set; the default build is restored on exit (build.py rebuilds when the flags differ).
out=$1; args=$2; shift 2
mkdir -p "$(dirname "$out")"
trap 'python3 -m salient_plusplus_amd.build > /dev/null 2>&1' EXIT
for fl in "$@"; do
  f="$fl"; [ "$fl" = "-" ] && f=""
  SPP_EXTRA_FLAGS="$f" python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build with '$fl' failed" >> "$out"; continue; }
  SPP_EXTRA_FLAGS="$f" timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags [$fl] ->', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'frac', round(d['roofline']['frac'],3))" >> "$out" 2>&1
done
cat "$out"
