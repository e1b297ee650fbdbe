"""Per-kernel SQ counter summary from a rocprofv3 --pmc CSV (counter_collection.csv).
usage: pmc_kernels.py <counter_collection.csv>
Prints, per spp:: kernel and grid size: launches, and per launch the average of every counter collected."""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "spp::" not in name:
        continue
    short = name.split("(")[0].replace("void ", "").replace("spp::", "")
    key = (short, int(r["Grid_Size"]) if "Grid_Size" in r else 0)
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    did = (r["Dispatch_Id"], key)
    if did not in seen:
        seen.add(did)
        n[key] += 1
names = sorted({c for v in acc.values() for c in v})
print(f"{'kernel':30s} {'grid':>9s} {'n':>4s} " + " ".join(f"{c[:16]:>16s}" for c in names))
for key in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], 0))):
    print(f"{key[0][:30]:30s} {key[1]:9d} {n[key]:4d} " + " ".join(f"{acc[key].get(c, 0) / n[key]:16.0f}" for c in names))
