"""Idle gaps between consecutive kernels of one sampling chain (kernel trace of `microbench.py chain` with ONE chain
in flight: SPP_WORK_STREAMS=1 CHAIN_CFG=8,8).  usage: chain_gaps.py <kernel_trace.csv>"""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "spp::" not in n:
        continue
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0].replace("void ", "").replace("spp::", "")))
rows.sort()
chains, cur = [], None
for s, e, n in rows:
    if n.startswith("k_seed_init"):
        if cur:
            chains.append(cur)
        cur = []
    if cur is not None:
        cur.append((s, e, n))
chains = [c for c in chains[5:-1] if len(c) >= 15]
tot = busy = gaps = 0
worst = {}
for c in chains:
    tot += c[-1][1] - c[0][0]
    busy += sum(e - s for s, e, n in c)
    for (s0, e0, n0), (s1, e1, n1) in zip(c, c[1:]):
        g = s1 - e0
        gaps += g
        worst[(n0, n1)] = worst.get((n0, n1), 0) + g
n = len(chains)
print(f"{n} chains: {tot / n / 1e3:.0f} us from first kernel start to last kernel end, {busy / n / 1e3:.0f} us inside kernels, "
      f"{gaps / n / 1e3:.0f} us in {len(chains[0]) - 1} gaps ({gaps / n / (len(chains[0]) - 1) / 1e3:.1f} us each)")
