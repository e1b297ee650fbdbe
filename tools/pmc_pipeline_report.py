"""Per-kernel L2<->fabric traffic of the whole pipeline from two rocprofv3 PMC passes (FETCH_SIZE,
WRITE_SIZE) over bench.py.   usage: pmc_pipeline_report.py fetch.csv write.csv n_delivered n_sampled [fetch_corr] [write_corr]
(n_delivered = batches the consumer took = k_deliver launches; n_sampled = batches the sampler produced,
i.e. launches of a per-hop kernel / hops * group size -- the sampler runs ahead of the consumer)
FETCH_SIZE counts half of the fetched bytes on gfx950 (MI355X_MICROARCH.md, HBM): default correction 2.0;
WRITE_SIZE is taken as read (the 8-B non-temporal stores of the gather read ~5 % high)."""
import collections
import csv
import sys

fetch_csv, write_csv, nb, ns = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
fc = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
wc = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0


def per_kernel(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "spp::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[name] += float(r["Counter_Value"]) * 1024.0      # KB -> B
        calls[name] += 1
    return tot, calls


f, calls = per_kernel(fetch_csv, "FETCH_SIZE")
w, _ = per_kernel(write_csv, "WRITE_SIZE")
print(f"{'kernel':36s} {'fetch MB/batch':>15s} {'write MB/batch':>15s} {'launches':>9s}")
tf = tw = 0.0
for name in sorted(set(f) | set(w), key=lambda k: -(f[k] * fc + w[k] * wc)):
    if calls[name] == 1:
        continue                                  # one-off set-up kernels (k_narrow_col)
    div = nb if "k_deliver" in name else ns
    a, b = f[name] * fc / div / 1e6, w[name] * wc / div / 1e6
    tf, tw = tf + a, tw + b
    print(f"{name:36s} {a:15.1f} {b:15.1f} {calls[name]:9d}")
print(f"{'TOTAL':36s} {tf:15.1f} {tw:15.1f}")
