#!/usr/bin/env python3
"""Attribute the wall time of one training step to kernels from a rocprofv3 kernel trace: every instant is
split equally among the kernels running at that instant (streams overlap, so summed durations exceed wall)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
sgd = [e for e in ev if e[2].startswith('sgd_kernel')]
t0, t1 = sgd[step][1], sgd[step + 1][1]
seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = sorted(set([e[0] for e in seg] + [e[1] for e in seg]))
import heapq
acc = collections.Counter(); cnt = collections.Counter(); raw = collections.Counter()
for s, e, n in seg:
    cnt[n] += 1; raw[n] += e - s
# sweep
events = []
for i, (s, e, n) in enumerate(seg):
    events.append((s, 1, i)); events.append((e, 0, i))
events.sort()
active = set(); last = None
for t, kind, i in events:
    if last is not None and active and t > last:
        share = (t - last) / len(active)
        for j in active:
            acc[seg[j][2]] += share
    last = t
    if kind: active.add(i)
    else: active.discard(i)
print("step wall %.2f ms" % ((t1 - t0) / 1e6))
for n, v in acc.most_common(25):
    print("%-64s n=%4d attributed %7.3f ms  summed %7.3f ms" % (n[:64], cnt[n], v / 1e6, raw[n] / 1e6))
