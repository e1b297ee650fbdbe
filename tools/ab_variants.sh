#!/bin/bash
# usage: tools/ab_variants.sh <outfile> <rounds> A B ...   (GPU box)
# Same-box A/B of whole sampler.hip variants kept under .ab/<name>.hip: build, bench, alternate.
out=$1; rounds=$2; shift 2
mkdir -p "$(dirname "$out")"
cp salient_plusplus_amd/csrc/sampler.hip /tmp/sampler_current.hip
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp ".ab/$v.hip" salient_plusplus_amd/csrc/sampler.hip
    python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build of $v failed" >> "$out"; continue; }
    p=$(timeout -k 10 300 python3 bench.py --steps 192 --warmup 16 --no-cpu-baseline --no-model-step 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms']*1e3,1))")
    c=$(CHAIN_CFG=16,8 WL=S-papers timeout -k 10 200 python3 tools/microbench.py chain 2>/dev/null | grep "chain only" | sed 's/.*batches, //')
    echo "round $r variant $v: pipeline ms_per_step/deliver_us $p ; chain alone $c" >> "$out"
  done
done
cp /tmp/sampler_current.hip salient_plusplus_amd/csrc/sampler.hip
cat "$out"
