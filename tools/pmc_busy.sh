#!/bin/bash
# usage: tools/pmc_busy.sh <outdir>   (GPU box): how busy the texture-address units, the vector L1s and the L2 are
# while the delivery kernel runs in the pipeline vs alone (one PMC pass each)
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
set1="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_BUSY_avr"
set2="GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RD_UNCACHED_32B_sum"
i=0
for set in "$set1" "$set2"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$out" -o "busy_pipe_$i" -- python3 bench.py --steps 96 --warmup 16 --no-cpu-baseline --no-model-step > "$out/pipe_$i.log" 2>&1 || { tail -3 "$out/pipe_$i.log"; continue; }
  F=128 STRIDE=256 TABLE_ROWS=111059956 ROWS=947000 timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$out" -o "busy_alone_$i" -- python3 tools/pmc_gather.py > "$out/alone_$i.log" 2>&1 || { tail -3 "$out/alone_$i.log"; continue; }
  for w in pipe alone; do f=$(find "$out" -name "busy_${w}_${i}_counter_collection.csv" | head -1); python3 tools/pmc_kernels.py "$f" > "$out/busy_${w}_$i.txt"; done
  rm -f "$out"/busy_*_counter_collection.csv "$out"/busy_*_agent_info.csv
done
cat "$out"/busy_*.txt
