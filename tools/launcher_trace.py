"""Who waits for whom on the host (SPP_TRACE_LAUNCHER=1 output of a bench run on stdin/file).
usage: launcher_trace.py <stderr log>
Per group g: when the launcher began / finished enqueuing its chain (L, l), when the consumer first asked for
it (N) and got it (G); prints the last groups: enqueue duration, how long before it was needed the enqueue
finished, and how long the consumer waited."""
import collections
import sys
ev = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    if not line.startswith("[spp trace]"):
        continue
    _, _, t, what, g = line.split()
    ev[int(g)].setdefault(what, int(t))
rows = [(g, e) for g, e in sorted(ev.items()) if all(k in e for k in "LlNG")]
print(f"{'group':>5s} {'enqueue us':>10s} {'enqueued before needed us':>25s} {'consumer waited us':>19s} {'needed at us':>13s}")
for g, e in rows[-24:]:
    print(f"{g:5d} {e['l'] - e['L']:10d} {e['N'] - e['l']:25d} {e['G'] - e['N']:19d} {e['N']:13d}")
