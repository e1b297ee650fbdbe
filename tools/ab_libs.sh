#!/bin/bash
# usage: tools/ab_libs.sh <outfile> <reps> "<bench args>" <lib.so> [<lib.so> ...]   (GPU box)
# Interleaved repetitions of PREBUILT library variants (built in the container with SPP_EXTRA_FLAGS and kept under .abt/):
# each is copied over salient_plusplus_amd/libspp_hip.so of the box's scratch copy, bench.py and the lone chain are run,
# the default library is put back at the end.  Prints min / median of ms_per_step and of the lone chain per variant.
out=$1; reps=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"
lib=salient_plusplus_amd/libspp_hip.so
cp "$lib" /tmp/libspp_default.so
trap 'cp /tmp/libspp_default.so "$lib"' EXIT
for r in $(seq 1 $reps); do
  k=0
  for v in "$@"; do
    k=$((k+1))
    cp "$v" "$lib" || continue
    ( timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg$k rep$r [$(basename $v)]', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1))"
      c=$(CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 300 python3 tools/microbench.py chain 2>/dev/null | grep "chain only" | sed 's/.*batches, //')
      echo "cfg$k rep$r [$(basename $v)] chain_alone $c" ) >> "$out" 2>&1
  done
done
python3 - "$out" <<'PY'
import re, sys, statistics as st
by, ch = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"(cfg\d+) rep\d+ \[(.*?)\] ms_per_step ([\d.]+)", line)
    if m:
        by.setdefault((m.group(1), m.group(2)), []).append(float(m.group(3)))
    m = re.match(r"(cfg\d+) rep\d+ \[(.*?)\] chain_alone ([\d.]+)", line)
    if m:
        ch.setdefault((m.group(1), m.group(2)), []).append(float(m.group(3)))
for k, v in by.items():
    c = ch.get(k, [0.0])
    print(f"{k[0]} [{k[1]}]: n={len(v)} ms_per_step min {min(v):.4f} median {st.median(v):.4f}  | chain alone min {min(c):.1f} median {st.median(c):.1f} us/batch")
PY
