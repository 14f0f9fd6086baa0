"""Print a rocprofv3 kernel_stats.csv summary: python tools/kstats.py <csv> [forwards | auto] [rows]
forwards = auto: the number of 2-image forwards in the trace is taken from the calls of tail_fuse64_kernel (4 per forward: one per output map)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
arg = sys.argv[2] if len(sys.argv) > 2 else "1"
if arg == "auto":
    nf = next((int(r["Calls"]) / 4.0 for r in rows if r["Name"].startswith("tail_fuse64_kernel")), 1.0)
else:
    nf = float(arg)
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"forwards in the trace: {nf:g}; total kernel ms per forward: {tot / nf / 1e6:.2f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print(f"{r['Name'][:52]:52s} calls/fwd {int(r['Calls']) / nf:7.1f} ms/fwd {int(r['TotalDurationNs']) / nf / 1e6:7.3f} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
