#!/bin/bash
# usage: tools/chain_trace.sh <outdir>   (GPU box): per-kernel times of the sampling-only run (S-papers, 32 slots, groups of 8)
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o chain -- python3 tools/microbench.py chain > "$out/chain_only.log" 2>&1 || { tail -5 "$out/chain_only.log"; exit 1; }
f=$(find "$out" -name "chain_kernel_trace.csv" | head -1); python3 tools/trace_report.py "$f" 256 ${CHAIN_GROUP:-16} > "$out/chain_only_trace_report.txt"; rm -f "$f" "$out"/chain_agent_info.csv
grep "chain only" "$out/chain_only.log" >> "$out/chain_only_trace_report.txt"
cat "$out/chain_only_trace_report.txt"
