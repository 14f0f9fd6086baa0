"""Print a rocprofv3 kernel_stats.csv summary: python tools/kstats.py <csv> [forwards]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms per forward: {tot / nf / 1e6:.2f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{r['Name'][:52]:52s} calls {r['Calls']:>5s} ms/fwd {int(r['TotalDurationNs']) / nf / 1e6:7.2f} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
