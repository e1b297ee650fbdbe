"""Everything the GPU ran around the longest idle gap of the delivery stream inside a window (kernel trace of bench.py --steps K).
usage: gap_dump.py <kernel_trace.csv> [K=20]"""
import csv
import sys

path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:60],
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
dl = [x for x in rows if "k_deliver" in x[3]]
dl = dl[-(6 * K):]                      # the last six windows
best = None
for a, b in zip(dl, dl[1:]):
    gap = b[0] - a[1]
    if gap < 300_000 and (best is None or gap > best[0]):   # not a window boundary (those idle longer and involve the host)
        best = (gap, a, b)
gap, a, b = best
print(f"longest in-window idle gap of the delivery stream: {gap / 1e3:.0f} us, between the delivery ending at 0 and the one starting at {gap / 1e3:.0f}")
t0 = a[1]
for s, e, q, n, g in rows:
    if e >= a[0] - 200_000 and s <= b[1] + 100_000:
        print(f"  queue {q}  {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  grid {g:7d}  {n}")
