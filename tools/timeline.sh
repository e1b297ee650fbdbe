#!/bin/bash
# usage: tools/timeline.sh <outdir> "<bench args>" (GPU box): kernel trace of bench.py -> queue timeline, group lag table, trace report, delivery gaps
out=$1; args=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
SPP_TRACE_LAUNCHER=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o q -- python3 bench.py $args --no-cpu-baseline --no-model-step > "$out/q.json" 2> "$out/q.err" || { tail -3 "$out/q.err"; exit 1; }
f=$(find "$out" -name "q_kernel_trace.csv" | head -1)
python3 tools/queue_timeline.py "$f" 12 > "$out/queue_timeline.txt"
python3 tools/group_lag.py "$f" 8 16 > "$out/group_lag.txt"
python3 tools/trace_report.py "$f" 128 > "$out/trace_report.txt"
python3 tools/launcher_trace.py "$out/q.err" > "$out/launcher.txt"
python3 - "$f" > "$out/deliver_gaps.txt" <<'PY'
import csv, sys
dl = []
for r in csv.DictReader(open(sys.argv[1])):
    if "k_deliver" in r["Kernel_Name"]:
        dl.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
dl.sort()
dl = dl[-160:]
print("delivery kernels (last 160): start offset us, duration us, idle gap before it us")
t0 = dl[0][0]
for k in range(1, len(dl)):
    print(f"{k:4d} {(dl[k][0]-t0)/1e3:9.1f} {(dl[k][1]-dl[k][0])/1e3:7.1f} {(dl[k][0]-dl[k-1][1])/1e3:7.1f}")
tot = dl[-1][1] - dl[0][0]
busy = sum(e - s for s, e in dl)
print(f"span {tot/1e3:.0f} us, busy {busy/1e3:.0f} us ({busy/tot:.0%}), per batch {tot/1e3/(len(dl)-1):.1f} us")
PY
rm -f "$f" "$out"/q_agent_info.csv
cat "$out/queue_timeline.txt"; tail -22 "$out/group_lag.txt"; tail -3 "$out/deliver_gaps.txt"
