#!/bin/bash
# usage: tools/overlap.sh <outdir> [sage|gat]      (GPU box)
# Kernel traces of the model-step leg alone and beside the data path + the per-kernel comparison.
out=$1; arch=${2:-sage}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
for leg in resident data; do
  LEG=$leg timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$out" -o "ov_$leg" -- python3 tools/overlap_trace.py $arch 48 > "$out/ov_$leg.log" 2>&1 || { tail -5 "$out/ov_$leg.log"; exit 1; }
  grep OVERLAP_TRACE "$out/ov_$leg.log"
done
r=$(find "$out" -name "ov_resident_kernel_trace.csv" | head -1); d=$(find "$out" -name "ov_data_kernel_trace.csv" | head -1)
(grep -h OVERLAP_TRACE "$out"/ov_*.log; python3 tools/overlap_report.py "$r" "$d" 32) > "$out/overlap_${arch}_${TAG:-base}.txt"
# keep only the compared steps of the two traces (the whole files are tens of MB)
rm -f "$r" "$d" "$out"/ov_*_agent_info.csv
cat "$out/overlap_${arch}_${TAG:-base}.txt" | cut -c1-200
