"""Per group of 8 batches: when its sampling chain started / ended and when its first / last delivery started
(kernel trace of bench.py).  usage: group_lag.py <kernel_trace.csv> [G=8] [last_n_groups=14]"""
import csv
import sys
path = sys.argv[1]
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
last = int(sys.argv[3]) if len(sys.argv) > 3 else 14
dl, seeds, ch = [], [], []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "spp::" not in n:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    if "k_deliver" in n:
        dl.append((s, e))
    else:
        ch.append((s, e, q, ("k_seed_init" in n or "k_hop0_fused" in n)))
dl.sort()
ch.sort()
# chains: per queue, a k_seed_init opens one; it ends with the last kernel before the queue's next k_seed_init
chains = []
open_by_q = {}
for s, e, q, is_seed in ch:
    if is_seed:
        if q in open_by_q:
            chains.append(open_by_q[q])
        open_by_q[q] = [s, e, q]
    elif q in open_by_q:
        open_by_q[q][1] = max(open_by_q[q][1], e)
chains += list(open_by_q.values())
chains.sort()
ng = min(len(chains), len(dl) // G)
t0 = dl[0][0]
print(f"{'group':>5s} {'queue':>5s} {'chain start':>12s} {'chain end':>10s} {'first delivery':>15s} {'last delivery':>14s} {'ready before needed (us)':>25s}")
for k in range(max(0, ng - last), ng):
    cs, ce, q = chains[k]
    f, l = dl[k * G][0], dl[k * G + G - 1][0]
    print(f"{k:5d} {q:>5s} {(cs - t0) / 1e3:12.0f} {(ce - t0) / 1e3:10.0f} {(f - t0) / 1e3:15.0f} {(l - t0) / 1e3:14.0f} {(f - ce) / 1e3:25.0f}")
