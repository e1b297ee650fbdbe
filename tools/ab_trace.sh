#!/bin/bash
# usage: tools/ab_trace.sh <outdir> <tag> [ENV=VAL ...]   (run on the GPU box)
# One profiled bench.py run (rocprofv3 --kernel-trace --stats) with the given environment; leaves
# <outdir>/<tag>_bench.json, <tag>_kernel_stats.csv and <tag>_trace_report.txt (steady-state timeline).
out=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 bench.py --no-cpu-baseline --no-model-step > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err" || exit 1
f=$(find "$out" -name "${tag}_kernel_trace.csv" | head -1)
python3 tools/trace_report.py "$f" 128 > "$out/${tag}_trace_report.txt"
rm -f "$f" "$out/${tag}_agent_info.csv" "$out/${tag}_domain_stats.csv"
python3 - "$out/${tag}_bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step", round(d["ms_per_step"], 4), "deliver_us", round(d["roofline"]["avg_launch_ms"] * 1e3, 1), "frac", round(d["roofline"]["frac"], 3))
PY
tail -4 "$out/${tag}_trace_report.txt"
