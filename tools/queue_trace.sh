#!/bin/bash
# usage: tools/queue_trace.sh <outdir> "<bench args>"   (GPU box)
out=$1; args=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o q -- python3 bench.py $args --no-cpu-baseline --no-model-step > "$out/q.json" 2> "$out/q.err" || { tail -3 "$out/q.err"; exit 1; }
f=$(find "$out" -name "q_kernel_trace.csv" | head -1); python3 tools/queue_timeline.py "$f" 12 > "$out/queue_timeline.txt"; rm -f "$f" "$out"/q_agent_info.csv
cat "$out/queue_timeline.txt"
