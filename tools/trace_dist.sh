#!/bin/bash
# usage: tools/trace_dist.sh <outdir>   (GPU box): steady-state kernel timeline of the partitioned path at world size 1
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$out" -o fd -- python3 bench.py --gpus 1 --force-distributed --no-cpu-baseline --no-model-step > "$out/fd.json" 2> "$out/fd.err" || { tail -5 "$out/fd.err"; exit 1; }
f=$(find "$out" -name "fd_kernel_trace.csv" | head -1)
python3 tools/trace_report.py "$f" 192 > "$out/fd_trace_report.txt"
rm -f "$f" "$out"/fd_agent_info.csv
cat "$out/fd_trace_report.txt"
