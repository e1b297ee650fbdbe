#!/bin/bash
# usage: tools/pmc_requests.sh <outdir>     (GPU box)
# L1->L2 request counts per kernel (reads / writes / atomics leaving the CUs' vector L1, L2 hits and misses):
# two PMC passes over the sampling-only run and over the row gather at papers shape.
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  CHAIN_CFG=${CHAIN_CFG:-64,16} WL=S-papers timeout -k 10 500 rocprofv3 --pmc $set --output-format csv -d "$out" -o "req_chain_$i" -- python3 tools/microbench.py chain > "$out/req_chain_$i.log" 2>&1 || { tail -5 "$out/req_chain_$i.log"; exit 1; }
  F=128 STRIDE=256 TABLE_ROWS=111059956 ROWS=947000 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out" -o "req_gather_$i" -- python3 tools/pmc_gather.py > "$out/req_gather_$i.log" 2>&1 || { tail -5 "$out/req_gather_$i.log"; exit 1; }
  f=$(find "$out" -name "req_chain_${i}_counter_collection.csv" | head -1); python3 tools/pmc_kernels.py "$f" > "$out/req_chain_$i.txt"
  f=$(find "$out" -name "req_gather_${i}_counter_collection.csv" | head -1); python3 tools/pmc_kernels.py "$f" > "$out/req_gather_$i.txt"
  rm -f "$out"/req_*_counter_collection.csv "$out"/req_*_agent_info.csv
done
cat "$out"/req_chain_*.txt "$out"/req_gather_*.txt
