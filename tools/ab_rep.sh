#!/bin/bash
# usage: tools/ab_rep.sh <outfile> <reps> "<bench args>" "ENV=VAL ..." ["ENV=VAL ..." ...]   (GPU box)
# Interleaved repetitions (run-to-run spread of bench.py on one box is ~+-2.5 %, larger than most single changes):
# every environment set is run <reps> times in turn; prints min / median ms_per_step per set.
# CHAIN=1 adds the sampling-only run (tools/microbench.py chain) per repetition.
out=$1; reps=$2; args=$3; shift 3
mkdir -p "$(dirname "$out")"
for r in $(seq 1 $reps); do
  k=0
  for envs in "$@"; do
    k=$((k+1))
    ( for kv in $envs; do export "$kv"; done
      timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d['windows']['ms_per_step_all']
print('cfg$k rep$r [$envs]', 'ms_per_step', round(d['ms_per_step'],4), 'deliver_us', round(d['roofline']['avg_launch_ms']*1e3,1), 'mean_window', round(sum(w)/len(w),4), 'median_window', round(d['windows']['ms_per_step_median'],4))"
      if [ -n "$CHAIN" ]; then
        c=$(CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 300 python3 tools/microbench.py chain 2>/dev/null | grep "chain only" | sed 's/.*batches, //')
        echo "cfg$k rep$r [$envs] chain_alone $c"
      fi ) >> "$out" 2>&1
  done
done
python3 - "$out" <<'PY'
import re, sys, statistics as st
by = {}
ch = {}
mw = {}
for line in open(sys.argv[1]):
    m = re.match(r"(cfg\d+) rep\d+ \[(.*?)\] ms_per_step ([\d.]+)", line)
    if m:
        by.setdefault((m.group(1), m.group(2)), []).append(float(m.group(3)))
        mm = re.search(r"mean_window ([\d.]+) median_window ([\d.]+)", line)
        if mm:
            mw.setdefault((m.group(1), m.group(2)), []).append((float(mm.group(1)), float(mm.group(2))))
    m = re.match(r"(cfg\d+) rep\d+ \[(.*?)\] chain_alone ([\d.]+)", line)
    if m:
        ch.setdefault((m.group(1), m.group(2)), []).append(float(m.group(3)))
for k, v in by.items():
    extra = ""
    if k in ch:
        extra = f"  chain alone min {min(ch[k]):.1f} median {st.median(ch[k]):.1f} us/batch"
    if k in mw:
        extra += f"  | windows: mean {st.median([x[0] for x in mw[k]]):.4f} median {st.median([x[1] for x in mw[k]]):.4f}"
    print(f"{k[0]} [{k[1]}]: n={len(v)} min {min(v):.4f} median {st.median(v):.4f} max {max(v):.4f}{extra}")
PY
