"""usage: trace_long.py <kernel_trace.csv> [ms=10]: kernels longer than `ms` and GPU-idle gaps longer than `ms`"""
import csv
import sys
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 10e6
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r["Queue_Id"])
               for r in csv.DictReader(open(sys.argv[1]))))
t0 = rows[0][0]
end = rows[0][1]
for s, e, n, q in rows:
    if e - s > thr:
        print(f"LONG KERNEL {(e - s) / 1e6:8.2f} ms at {(s - t0) / 1e6:10.2f} ms  q{q} {n}")
    if s - end > thr:
        print(f"IDLE GAP    {(s - end) / 1e6:8.2f} ms at {(end - t0) / 1e6:10.2f} ms  (next: q{q} {n})")
    end = max(end, e)
