#!/bin/bash
# usage: tools/ab_variants_parts.sh <outfile> <rounds> A B ...  (GPU box): sampling-only run WITH ownership bucketing (rank 0 of 8, cache map)
out=$1; rounds=$2; shift 2
mkdir -p "$(dirname "$out")"
cp salient_plusplus_amd/csrc/sampler.hip /tmp/sampler_current.hip
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    cp ".ab/$v.hip" salient_plusplus_amd/csrc/sampler.hip
    python3 -m salient_plusplus_amd.build > /dev/null 2>&1 || { echo "build of $v failed" >> "$out"; continue; }
    c=$(CHAIN_PARTS=8 CHAIN_CFG=16,8 WL=S-papers timeout -k 10 300 python3 tools/microbench.py chain 2>&1 | grep "chain only" | sed 's/.*batches, //')
    echo "round $r variant $v: chain alone with 8-way bucketing $c" >> "$out"
  done
done
cp /tmp/sampler_current.hip salient_plusplus_amd/csrc/sampler.hip
cat "$out"
