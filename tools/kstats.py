"""usage: kstats.py <kernel_stats.csv> <steps> [min_calls]: per-step kernel time table from rocprofv3 --stats"""
import csv
import sys
steps = float(sys.argv[2])
minc = int(sys.argv[3]) if len(sys.argv) > 3 else int(steps)
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) >= minc]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    print(f"{per:8.1f} us/step {int(r['Calls']) / steps:5.1f}/step {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:120]}")
print(f"{tot:8.1f} us/step total kernel time")
