#!/bin/bash
# usage: tools/ab_chain.sh <outfile> "ENV=VAL ..." ["ENV=VAL ..."]   (GPU box): sampling-only throughput per environment
out=$1; shift
mkdir -p "$(dirname "$out")"
for envs in "$@"; do
  ( for kv in $envs; do export "$kv"; done
    echo -n "$envs -> "; CHAIN_CFG=${CHAIN_CFG:-16,8} WL=S-papers timeout -k 10 300 python3 tools/microbench.py chain 2>/dev/null | grep "chain only" ) >> "$out" 2>&1
done
cat "$out"
