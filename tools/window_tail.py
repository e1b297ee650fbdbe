"""Where a K-step window of bench.py spends its time (kernel trace of `bench.py --steps K`).
usage: window_tail.py <kernel_trace.csv> [K=20]
Deliveries come in bursts of K (one window each, closed by a synchronize).  Per window: the span of its K
delivery kernels, and the tail from the end of the last delivery to the end of the last sampling kernel that
started before the next window's first delivery (= what the closing synchronize waited for)."""
import csv
import sys

path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dl, ch = [], []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "spp::" not in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    (dl if "k_deliver" in n else ch).append((s, e))
dl.sort()
ch.sort()
# The trace also holds the priming and warm-up deliveries (3 x slots + W: not a multiple of K), so the windows do not start at
# delivery 0: a window boundary is where the closing synchronize sits, i.e. the longest gaps between consecutive deliveries --
# take the offset (mod K) whose boundaries carry the largest total gap over the last windows.  (Round 4 cut the windows at
# offset 0 and read the synchronize as "a gap before the last two deliveries".)
tail = dl[-min(len(dl), 12 * K):]
best, off = -1, 0
for o in range(K):
    tot = sum(tail[i][0] - tail[i - 1][1] for i in range(1, len(tail)) if (len(dl) - len(tail) + i) % K == o)
    if tot > best:
        best, off = tot, o
print(f"window boundaries at delivery index = {off} mod {K}")
dl = dl[off:]
wins = [dl[i:i + K] for i in range(0, len(dl) - K + 1, K)]
rows = []
for w, nxt in zip(wins[-9:-1], wins[-8:]):
    t0, t_last = w[0][0], w[-1][1]
    t_next = nxt[0][0]
    tail_end = max([e for s, e in ch if s < t_next and e > t_last] + [t_last])
    rows.append(((t_last - t0) / 1e3, (tail_end - t_last) / 1e3, (t_next - t0) / 1e3))
print(f"{'deliveries span us':>20s} {'chain tail us':>14s} {'window (to next) us':>20s}")
for a, b, c in rows:
    print(f"{a:20.0f} {b:14.0f} {c:20.0f}")

# detail of the last complete window: every delivery (start offset, duration) and the sampling kernels' busy spans
if len(wins) >= 2:
    w, nxt = wins[-2], wins[-1]
    t0, t_next = w[0][0], nxt[0][0]
    print("\nlast complete window, delivery kernels (start us, duration us):")
    print("  " + "  ".join(f"{(s - t0) / 1e3:.0f}+{(e - s) / 1e3:.0f}" for s, e in w))
    spans = []
    for s, e in ch:
        if e < t0 or s > t_next:
            continue
        if spans and s <= spans[-1][1] + 2000:
            spans[-1][1] = max(spans[-1][1], e)
        else:
            spans.append([s, e])
    print("sampling kernels busy (start us .. end us, gaps < 2 us merged):")
    print("  " + "  ".join(f"{(s - t0) / 1e3:.0f}..{(e - t0) / 1e3:.0f}" for s, e in spans))

# chain starts (k_seed_init launches) and ends (last k_hop_rows / k_gpart_scatter before the next start) per queue
if len(wins) >= 3:
    w0, w1 = wins[-3], wins[-1]
    ta, tb = w0[0][0], w1[-1][1]
    t0 = wins[-2][0][0]
    ev = []
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "spp::" not in n or "k_deliver" in n:
            continue
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e < ta or s > tb:
            continue
        ev.append((s, e, r.get("Queue_Id", "?"), n.split("(")[0].replace("void ", "").replace("spp::", "")))
    ev.sort()
    print("\nchains around the last complete window (queue: start .. end us relative to its first delivery; one per k_seed_init):")
    by_q = {}
    for s, e, q, n in ev:
        by_q.setdefault(q, []).append((s, e, n))
    for q, lst in by_q.items():
        cur = None
        out = []
        for s, e, n in lst:
            if n.startswith("k_seed_init"):
                if cur:
                    out.append(cur)
                cur = [s, e]
            elif cur:
                cur[1] = max(cur[1], e)
        if cur:
            out.append(cur)
        print(f"  queue {q}: " + "  ".join(f"{(s - t0) / 1e3:.0f}..{(e - t0) / 1e3:.0f}" for s, e in out))
    print("deliveries of the three windows (start us): " + " ".join(f"{(s - t0) / 1e3:.0f}" for s, e in wins[-3] + wins[-2] + wins[-1]))
