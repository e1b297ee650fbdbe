"""Per hardware queue: the sampling chains' (start..end) spans over the last N ms of a bench.py kernel trace, and
how busy each queue was.  usage: queue_timeline.py <kernel_trace.csv> [ms=12]"""
import csv
import sys
path = sys.argv[1]
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "spp::" not in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n.split("(")[0].replace("void ", "").replace("spp::", "")))
rows.sort()
dl = [x for x in rows if x[3].startswith("k_deliver")]
t_end = dl[-1][1]
t_beg = t_end - int(ms * 1e6)
by_q = {}
for s, e, q, n in rows:
    if e < t_beg or s > t_end:
        continue
    by_q.setdefault(q, []).append((s, e, n))
for q, lst in sorted(by_q.items()):
    busy = sum(e - s for s, e, n in lst)
    if lst[0][2].startswith("k_deliver"):
        print(f"queue {q} (delivery): {len(lst)} kernels, busy {busy / (t_end - t_beg):.0%}")
        continue
    chains, cur = [], None
    for s, e, n in lst:
        if (n.startswith("k_seed_init") or n.startswith("k_hop0_fused")):
            if cur:
                chains.append(cur)
            cur = [s, e]
        elif cur:
            cur[1] = max(cur[1], e)
    if cur:
        chains.append(cur)
    print(f"queue {q}: busy {busy / (t_end - t_beg):.0%}; chains (start..end us): " +
          "  ".join(f"{(s - t_beg) / 1e3:.0f}..{(e - t_beg) / 1e3:.0f}" for s, e in chains))

# delivery stalls: gaps > 250 us between consecutive delivery starts, and the chain that ended inside each
allch = []
for q, lst in by_q.items():
    if lst[0][2].startswith("k_deliver"):
        continue
    cur = None
    for s, e, n in lst:
        if (n.startswith("k_seed_init") or n.startswith("k_hop0_fused")):
            if cur:
                allch.append(cur)
            cur = [s, e, q]
        elif cur:
            cur[1] = max(cur[1], e)
    if cur:
        allch.append(cur)
d2 = [x for x in dl if x[0] >= t_beg]
print("\ndelivery stalls (> 250 us between starts): gap start/end us, index of the resuming delivery, chains ending in the gap (queue: start..end)")
for k in range(1, len(d2)):
    if d2[k][0] - d2[k - 1][0] > 250000:
        g0, g1 = d2[k - 1][1], d2[k][0]
        inside = [c for c in allch if g0 - 50000 <= c[1] <= g1]
        print(f"  {(g0 - t_beg) / 1e3:.0f}..{(g1 - t_beg) / 1e3:.0f} (#{len(dl) - len(d2) + k}): " +
              "  ".join(f"q{c[2]}: {(c[0] - t_beg) / 1e3:.0f}..{(c[1] - t_beg) / 1e3:.0f}" for c in inside))
