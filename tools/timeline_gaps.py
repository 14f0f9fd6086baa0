"""GPU busy / idle time of the replayed step from a rocprofv3 kernel trace: union of the kernel intervals against wall time, and the
time with exactly 1 / 2+ kernels in flight.  python tools/timeline_gaps.py <kernel_trace.csv> [steps=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the timed region = the last `steps` replays: take the last 60 % of the dispatches' time span as a steady-state window
t_end = max(e for _, e, _ in ev)
n = len(ev)
win = ev[int(n * 0.45):]            # steady state: after warm-up / capture, before the closing roofline pass is irrelevant (run with --no-roofline)
t0, t1 = win[0][0], max(e for _, e, _ in win)
pts = []
for s, e, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = {0: 0, 1: 0, 2: 0}
cur, last = 0, pts[0][0]
for t, d in pts:
    busy[min(cur, 2)] += t - last
    cur += d; last = t
tot = t1 - t0
print(f"window {tot / 1e6:.1f} ms, {len(win)} dispatches: idle {busy[0] / tot * 100:.1f} %, one kernel {busy[1] / tot * 100:.1f} %, two or more {busy[2] / tot * 100:.1f} %")
gaps = []
cur, last = 0, pts[0][0]
for t, d in pts:
    if cur == 0 and t > last:
        gaps.append(t - last)
    cur += d; last = t
gaps.sort()
if gaps:
    print(f"idle gaps: {len(gaps)}, median {gaps[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, total {sum(gaps) / 1e6:.2f} ms")
