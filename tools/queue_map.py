"""Which hardware queue every spp kernel ran on (kernel trace CSV).  usage: queue_map.py <kernel_trace.csv>"""
import collections
import csv
import sys
cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "").replace("spp::", "")[:28] if "spp::" in n else "(other)"
    cnt[r.get("Queue_Id", "?")][short] += 1
for q in sorted(cnt):
    print(f"queue {q}: " + ", ".join(f"{k} x{v}" for k, v in cnt[q].most_common(12)))
