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
