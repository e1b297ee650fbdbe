#!/bin/bash
# round 5: single-stream kernel trace of the lone chain, old (flag + rows) vs new (finish)
OUT=${1:-gpurun_out/r5k}; mkdir -p $OUT
for fin in 0 1; do
  SPP_FINISH=$fin SPP_WORK_STREAMS=1 bash tools/chain_trace.sh $OUT/fin$fin > /dev/null 2>&1
  echo "=== SPP_FINISH=$fin (one stream)"; cat $OUT/fin$fin/chain_only_trace_report.txt
done
