#!/bin/bash
# usage: tools/chain_trace_parts.sh <outdir>   (GPU box): per-kernel times of the sampling-only run with 8-way ownership bucketing
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
CHAIN_PARTS=8 CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o chain -- python3 tools/microbench.py chain > "$out/chain_only.log" 2>&1 || { tail -5 "$out/chain_only.log"; exit 1; }
f=$(find "$out" -name "chain_kernel_trace.csv" | head -1); python3 tools/trace_report.py "$f" 256 8 > "$out/chain_parts_trace_report.txt"; rm -f "$f" "$out"/chain_agent_info.csv
grep "chain only" "$out/chain_only.log" >> "$out/chain_parts_trace_report.txt"
cat "$out/chain_parts_trace_report.txt"
