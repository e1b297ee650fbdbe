"""Concurrency analysis of a rocprofv3 kernel trace: wall span, union-busy time, per-queue busy."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'].split('(')[0][-40:]) for r in rows]
ev.sort()
# steady state: last 60% of the spp kernels
spp = [e for e in ev if 'spp::' in e[3]]
t_lo = spp[int(len(spp) * 0.4)][0]
t_hi = spp[-1][1]
win = [e for e in ev if e[0] >= t_lo and e[1] <= t_hi]
span = (t_hi - t_lo) / 1e3
busy = 0; cur_s, cur_e = None, None
for s, e, q, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, q, n in win)
print(f"window {span:.0f} us, union busy {busy/1e3:.0f} us ({busy/1e3/span:.1%}), sum of kernel time {tot/1e3:.0f} us (avg concurrency {tot/busy:.2f})")
perq = collections.Counter()
for s, e, q, n in win: perq[q] += e - s
print("per queue busy us:", {q: round(v/1e3) for q, v in perq.items()})
nb = sum(1 for e in win if 'k_seed_init' in e[3])
print(f"batches in window: {nb}, => {span/max(nb,1):.0f} us per batch")
pern = collections.Counter()
for s, e, q, n in win: pern[n] += e - s
for n, v in pern.most_common(12): print(f"  {n:42s} {v/1e3/max(nb,1):8.1f} us/batch")
