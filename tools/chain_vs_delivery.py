"""Which delivery was running when each sampling chain started / ended, and where the delivery stream stood idle
(kernel trace of bench.py).  usage: chain_vs_delivery.py <kernel_trace.csv> [first delivery shown = -120] [count = 120]"""
import csv
import sys

path = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -120
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
dl, seeds, ends = [], [], []
last_by_q = {}
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "spp::" not in n:
        continue
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    if "k_deliver" in n:
        dl.append((s, e))
    else:
        if "k_seed_init" in n:
            seeds.append((s, q))
        last_by_q.setdefault(q, []).append((s, e, n.split("(")[0].replace("void ", "").replace("spp::", "")))
dl.sort()
seeds.sort()
# a chain = the kernels of one queue from a k_seed_init to the next one on that queue
chains = []
for q, ks in last_by_q.items():
    ks.sort()
    cur = None
    for s, e, n in ks:
        if n.startswith("k_seed_init"):
            if cur:
                chains.append(cur)
            cur = [s, e, q]
        elif cur:
            cur[1] = max(cur[1], e)
    if cur:
        chains.append(cur)
chains.sort()
lo = len(dl) + first if first < 0 else first
print(f"{len(dl)} deliveries, {len(chains)} chains, {len(seeds)} k_seed_init launches")
t0 = dl[lo][0]
ev = []
for i in range(lo, min(len(dl), lo + count)):
    gap = dl[i][0] - dl[i - 1][1] if i > 0 else 0
    ev.append((dl[i][0], f"delivery {i:5d} start {(dl[i][0] - t0) / 1e3:9.0f} us  dur {(dl[i][1] - dl[i][0]) / 1e3:5.0f}" +
               (f"   <-- delivery stream idle {gap / 1e3:.0f} us before" if gap > 30000 else "")))
for s, e, q in chains:
    if dl[lo][0] - 3_000_000 <= s <= dl[min(len(dl) - 1, lo + count - 1)][1]:
        ev.append((s, f"    chain start (queue {q})  {(s - t0) / 1e3:9.0f} us"))
        ev.append((e, f"    chain end   (queue {q})  {(e - t0) / 1e3:9.0f} us  (took {(e - s) / 1e3:.0f} us)"))
for _, line in sorted(ev):
    print(line)
