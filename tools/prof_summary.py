import csv, sys, collections
p = sys.argv[1]
rows = list(csv.DictReader(open(p)))
print(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'pct':>6s} {'max_us':>9s}")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 18]:
    name = r['Name'].split('(')[0][-58:]
    print(f"{name:60s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f} {float(r['MaxNs'])/1e3:9.1f}")
