#!/bin/bash
# usage: tools/window_trace.sh <outdir> [K=20]   (GPU box)
out=$1; K=${2:-20}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o w -- python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-model-step > "$out/w.json" 2> "$out/w.err" || { tail -3 "$out/w.err"; exit 1; }
f=$(find "$out" -name "w_kernel_trace.csv" | head -1); python3 tools/window_tail.py "$f" $K > "$out/window_tail.txt"; python3 tools/group_lag.py "$f" 8 16 > "$out/group_lag.txt"; python3 tools/queue_map.py "$f" > "$out/queue_map.txt"; python3 tools/gap_dump.py "$f" $K > "$out/gap_dump.txt"; rm -f "$f" "$out"/w_agent_info.csv
cat "$out/queue_map.txt"
