"""Concurrency analysis of one graph-replayed step from a rocprofv3 kernel trace: how much wall time has 1, 2, 3+ kernels
running, and which kernels own the time during which they run ALONE (the critical path's exposed time)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:44]) for r in rows)
segs, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-60:]) > 300000:
        segs.append(cur); cur = [e]
    else:
        cur.append(e)
segs.append(cur)
want = int(sys.argv[2]) if len(sys.argv) > 2 else collections.Counter(len(s) for s in segs if len(s) > 100).most_common(1)[0][0]
seg = [s for s in segs if len(s) == want][-1]
t0, t1 = seg[0][0], max(x[1] for x in seg)
pts = []
for s, e, n in seg:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort()
active = collections.Counter(); last = t0
by_level = collections.Counter(); alone = collections.Counter(); share = collections.Counter()
for t, d, n in pts:
    dt = t - last
    k = sum(active.values())
    if dt > 0:
        by_level[min(k, 4)] += dt
        if k == 1:
            alone[next(iter(x for x in active if active[x] > 0))] += dt
        for x in active:
            if active[x] > 0:
                share[x] += dt / k
    active[n] += d; last = t
tot = t1 - t0
print(f"step wall {tot/1e6:.2f} ms, kernels {len(seg)}, sum of durations {sum(e-s for s,e,_ in seg)/1e6:.2f} ms")
print("time with k kernels running: " + ", ".join(f"{k}{'+' if k==4 else ''}: {v/1e6:.2f} ms" for k, v in sorted(by_level.items())))
print("attributed wall time (time / #concurrent kernels), top 16:")
for n, v in share.most_common(16):
    print(f"  {n:44s} {v/1e6:6.2f} ms   alone {alone[n]/1e6:6.2f} ms")
