#!/bin/bash
# usage: tools/ab_flags_rep.sh <outfile> <reps> "<bench args>" "<flags>|<ENV=VAL ...>" ...   (GPU box; "-" = default flags)
# Interleaved repetitions of (compile flags, environment) pairs; prints min / median ms_per_step per pair.
out=$1; reps=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"
trap 'python3 -m salient_plusplus_amd.build > /dev/null 2>&1' EXIT
for r in $(seq 1 $reps); do
  k=0
  for spec in "$@"; do
    k=$((k+1))
    fl="${spec%%|*}"; envs="${spec#*|}"; [ "$envs" = "$spec" ] && envs=""
    f="$fl"; [ "$fl" = "-" ] && f=""
    SPP_EXTRA_FLAGS="$f" python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build with '$fl' failed" >> "$out"; continue; }
    ( for kv in $envs; do export "$kv"; done
      SPP_EXTRA_FLAGS="$f" timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg$k rep$r [$spec]', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1))" ) >> "$out" 2>&1
  done
done
python3 - "$out" <<'PY'
import re, sys, statistics as st
by = {}
for line in open(sys.argv[1]):
    m = re.match(r"(cfg\d+) rep\d+ \[(.*?)\] ms_per_step ([\d.]+) deliver_us ([\d.]+)", line)
    if m:
        by.setdefault((m.group(1), m.group(2)), []).append((float(m.group(3)), float(m.group(4))))
for k, v in by.items():
    a = [x[0] for x in v]; b = [x[1] for x in v]
    print(f"{k[0]} [{k[1]}]: n={len(a)} min {min(a):.4f} median {st.median(a):.4f} max {max(a):.4f}  deliver median {st.median(b):.0f} us")
PY
