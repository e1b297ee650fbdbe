"""Consumer-side cadence from a rocprofv3 rocpd database: start-to-start deltas of the per-batch
delivery kernel in the last epoch, and the host API calls of the consumer thread around the longest
GPU-idle gap (development aid; run on the GPU box: the database is too large to copy back)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 192
k = sorted(c.execute("select start,end,name from kernels").fetchall())
d = [(s, e) for s, e, n in k if 'k_deliver' in n][-nlast:]
deltas = [round((d[i + 1][0] - d[i][0]) / 1e3) for i in range(len(d) - 1)]
print("deliver start-to-start deltas (us), last epoch:")
for i in range(0, len(deltas), 16):
    print("  ", deltas[i:i + 16])
lo, hi = d[0][0], d[-1][1]
cur_end = None
gaps = []
for s, e, n in k:
    if s < lo or s > hi:
        continue
    if cur_end is not None and s > cur_end:
        gaps.append((s - cur_end, cur_end, s))
    cur_end = max(cur_end or e, e)
gaps.sort(reverse=True)
print("largest GPU-idle gaps (us):", [round(g[0] / 1e3) for g in gaps[:12]])
if not gaps:
    sys.exit(0)
g = gaps[0]
a, b = g[1] - 100000, g[2] + 50000
rows = c.execute("select start,end,tid,name from regions where end>=? and start<=? order by start", (a, b)).fetchall()
tids = {}
for s, e, tid, n in rows:
    tids[tid] = tids.get(tid, 0) + 1
print("threads (calls in window):", tids)
# consumer thread = the one that calls hipEventSynchronize / hipLaunchKernel least like a poller: print all but pure pollers
poll = {'hipEventQuery', 'hipGetDevice', 'hipSetDevice', 'hipThreadExchangeStreamCaptureMode', 'hipGetLastError',
        'hipPeekAtLastError'}
print(f"gap of {g[0]/1e3:.0f} us at [100, {(g[2]-a)/1e3:.0f}] us of this window:")
last = {}
for s, e, tid, n in rows:
    if n in poll:
        continue
    idle = (s - last.get(tid, s)) / 1e3
    last[tid] = e
    print(f"{(s-a)/1e3:9.1f} {(e-s)/1e3:8.1f} tid{tid % 1000:03d} idle_before={idle:7.1f} {n}")
