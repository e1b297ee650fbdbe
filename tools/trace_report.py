"""Steady-state timeline report from a rocprofv3 --kernel-trace CSV of bench.py.

usage: trace_report.py <kernel_trace.csv> [n_last_batches=128] [batches per group = 8]

Window = the delivery launches of the last n batches (a k_deliver_group launch delivers a whole group).  Prints, for the spp:: kernels inside it: per kernel (and per
grid size, which tells the hops apart) calls / total / average duration; per-batch kernel time of the
delivery kernel and of the sampling chain; the time at least one kernel was running (union), the sum
of kernel time, the average concurrency and the window per batch."""
import collections
import csv
import sys

path = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 128
rows = []
for r in csv.DictReader(open(path)):
    name = r["Kernel_Name"]
    if "spp::" not in name:
        continue
    short = name.split("(")[0].replace("void ", "").replace("spp::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short,
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), r["Queue_Id"]))
rows.sort()
deliver = [x for x in rows if x[2].startswith("k_deliver")]
per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 8     # batches per sampling / delivery group
if not deliver:
    # sampling only (tools/microbench.py chain): the window is the last n batches' worth of k_seed_init launches
    seeds = [x for x in rows if (x[2].startswith("k_seed_init") or x[2].startswith("k_hop0_fused"))]
    k = max(1, min(len(seeds) - 1, nlast // per_launch))
    lo, hi = seeds[-k - 1][0], seeds[-1][0]
    nlast = k * per_launch
else:
    got, first = 0, len(deliver)
    while first > 0 and got < nlast:
        first -= 1
        got += per_launch if deliver[first][2].startswith("k_deliver_group") else 1
    nlast = got
    lo, hi = deliver[first][0], deliver[-1][1]
win = [x for x in rows if x[0] >= lo and x[1] <= hi]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n, gx, gy, q in win:
    key = (n, gx, gy)
    tot[key][0] += 1
    tot[key][1] += e - s
print(f"window: last {nlast} batches, {(hi - lo) / 1e3:.0f} us = {(hi - lo) / 1e3 / nlast:.1f} us per batch")
print(f"{'kernel':34s} {'grid':>14s} {'calls':>6s} {'total_us':>10s} {'avg_us':>8s} {'us/batch':>9s}")
by_name = collections.defaultdict(int)
for (n, gx, gy), (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    by_name[n] += t
    print(f"{n:34s} {f'{gx}x{gy}':>14s} {c:6d} {t / 1e3:10.1f} {t / 1e3 / c:8.1f} {t / 1e3 / nlast:9.2f}")
print("\nper kernel, all grid sizes (us per batch):")
for n, t in sorted(by_name.items(), key=lambda kv: -kv[1]):
    print(f"  {n:34s} {t / 1e3 / nlast:8.2f}")
chain = sum(t for n, t in by_name.items() if not n.startswith("k_deliver"))
dl = sum(t for n, t in by_name.items() if n.startswith("k_deliver"))
ev = sorted([(s, 1) for s, e, *_ in win] + [(e, -1) for s, e, *_ in win])
busy = 0
depth = 0
last = None
for t, d in ev:
    if depth > 0:
        busy += t - last
    depth += d
    last = t
ksum = chain + dl
print(f"\nkernel time per batch: delivery {dl / 1e3 / nlast:.1f} us, sampling chain {chain / 1e3 / nlast:.1f} us, "
      f"sum {ksum / 1e3 / nlast:.1f} us")
print(f"GPU busy (>= 1 kernel running) {busy / (hi - lo):.1%} of the window; average concurrency while busy "
      f"{ksum / max(busy, 1):.2f}")
