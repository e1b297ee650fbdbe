"""Per-kernel figures of the exchange path at the 8-rank composition (tools/exchange_p8.py under rocprofv3).
usage: exchange_p8_report.py <kernel_trace.csv> <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> <exchange_p8 stdout>
Traffic = L2<->fabric bytes from the two PMC passes (FETCH_SIZE x 2: it counts 64-byte units as 32 on gfx950,
MI355X_MICROARCH.md; WRITE_SIZE as read), summed over ALL ranks' launches and divided by the rows all ranks moved
through that kernel.  Durations are from the kernel trace of the same command: eight ranks share one GPU there, so a
kernel's duration is an upper bound of what a rank alone on its GPU sees."""
import collections
import csv
import json
import sys

trace, fcsv, wcsv, log = sys.argv[1:5]
summ = None
for ln in open(log):
    if ln.startswith("EXCHANGE_P8 "):
        summ = json.loads(ln[len("EXCHANGE_P8 "):])
if summ is None:
    raise SystemExit("no EXCHANGE_P8 line in " + log)


def short(n):
    return n.split("(")[0].replace("void ", "").replace("spp::", "")


dur = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(trace)):
    if "spp::" in r["Kernel_Name"]:
        d = dur[short(r["Kernel_Name"])]
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def pmc(path, counter):
    tot = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "spp::" in r["Kernel_Name"]:
            tot[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024.0
    return tot


f, w = pmc(fcsv, "FETCH_SIZE"), pmc(wcsv, "WRITE_SIZE")
rb = summ["row_bytes"]
units = {  # kernel prefix -> (rows all ranks moved through it, algorithmic bytes per row, what a row is)
    "k_serve_rows": (summ["rows_served"], 2 * rb + 4, "served row: int32 id + row read + row written"),
    "k_deliver": (summ["rows_delivered"], 2 * rb + 12, "assembled row: {bucket, row} record + int64 n_id written + row read + row written"),
    "k_pack_remote_ids": (summ["rows_fetched"], 8, "requested id: int32 read + int32 written"),
    "k_gpart_hist": (summ["rows_delivered"], 5, "node: int32 id read + bucket byte written"),
    "k_gpart_scatter": (summ["rows_delivered"], 4 + 1 + 4 + 4 + 8, "node: id + bucket read, perm + parts + {bucket,row} written"),
}
print(f"workload {summ['workload']}, P = {summ['P']} in-process ranks on one GPU, {summ['batches_all_ranks']} batches over all ranks: "
      f"{summ['rows_delivered'] / summ['batches_all_ranks'] / 1e3:.0f} k rows per batch = {100 * summ['frac_local']:.1f} % local, "
      f"{100 * summ['frac_cache']:.1f} % cache hits, {100 * summ['frac_fetched']:.1f} % received from peers; row = {rb} B")
print(f"\n{'kernel':28s} {'launches':>8s} {'avg us (8 ranks share the GPU)':>31s} {'fetch MB':>10s} {'write MB':>10s} {'rows (all ranks)':>17s} "
      f"{'traffic B/row':>14s} {'algorithmic B/row':>18s}")
for name in sorted(dur, key=lambda k: -dur[k][1]):
    n, ns = dur[name]
    fb, wb = 2.0 * f.get(name, 0.0), w.get(name, 0.0)
    u = next((v for k, v in units.items() if name.startswith(k)), None)
    per = f"{(fb + wb) / u[0]:14.1f} {u[1]:18d}" if u and u[0] else f"{'':14s} {'':18s}"
    rows = f"{u[0]:17d}" if u else f"{'':17s}"
    print(f"{name:28s} {n:8d} {ns / n / 1e3:31.1f} {fb / 1e6:10.1f} {wb / 1e6:10.1f} {rows} {per}")
print("\nrow meanings: " + "; ".join(f"{k}: {v[2]}" for k, v in units.items()))
