"""usage: api_long.py <hip_api_trace.csv> [ms=10]: HIP API calls longer than `ms` (which call did a host stall sit in?)"""
import csv
import sys
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > thr:
        print(f"{d / 1e6:8.2f} ms at {(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  tid {r.get('Thread_Id')}  {r['Function']}")
