#!/usr/bin/env python3
"""Which kernels differ between the fast and the slow steps of ONE rocprofv3 kernel trace (tools only).
Steps are cut at sgd_kernel; a step is "slow" when its wall time is above the midpoint of the fastest and slowest step.
usage: mode_switch_diff.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:60],
             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), r.get('Queue_Id', '')) for r in rows)
sgd = [e for e in ev if e[2].startswith('sgd_kernel')]
steps = []
for a, b in zip(sgd[:-1], sgd[1:]):
    seg = [e for e in ev if e[0] >= a[1] and e[1] <= b[1]]
    steps.append(((b[1] - a[1]) / 1e6, seg))
walls = [w for w, _ in steps]
print("step walls (ms):", " ".join("%.1f" % w for w in walls))
lo, hi = min(walls[2:]), max(walls[2:])
mid = (lo + hi) / 2
fast = [s for w, s in steps[2:] if w <= mid]; slow = [s for w, s in steps[2:] if w > mid]
print("fast steps %d (%.2f ms), slow steps %d (%.2f ms)" % (len(fast), sum(w for w in walls[2:] if w <= mid) / max(1, len(fast)),
                                                              len(slow), sum(w for w in walls[2:] if w > mid) / max(1, len(slow))))
def agg(group):
    g = collections.defaultdict(float); q = collections.defaultdict(float)
    for seg in group:
        for s, e, n, wgs, qu in seg:
            g[(n, wgs)] += (e - s) / 1e6 / len(group); q[qu] += (e - s) / 1e6 / len(group)
    return g, q
if fast and slow:
    gf, qf = agg(fast); gs, qs = agg(slow)
    print("per queue busy ms  fast / slow:", {k: (round(qf[k], 2), round(qs.get(k, 0), 2)) for k in qf})
    d = sorted(((gs.get(k, 0) - gf.get(k, 0), k) for k in set(gf) | set(gs)), reverse=True)
    print("%-62s %8s %9s %9s %9s" % ("kernel", "wgs", "fast ms", "slow ms", "diff"))
    for diff, k in d[:25] + d[-8:]:
        print("%-62s %8d %9.3f %9.3f %+9.3f" % (k[0], k[1], gf.get(k, 0), gs.get(k, 0), diff))
# time line of the mask head's backward pass (mask_out_bwd .. its ROIAlign adjoint) in one fast and one slow step
def window(seg, title):
    a = next((e for e in seg if 'mask_out_bwd' in e[2]), None)
    b = next((e for e in reversed(seg) if e[2].startswith('roialign_kernel<true>')), None)
    if not a or not b: return
    print("\n%s: mask-head backward window %.3f ms" % (title, (b[1] - a[0]) / 1e6))
    for s, e, n, wgs, qu in seg:
        if s >= a[0] - 200000 and s <= b[1] and (e - s) > 30000:
            print("  q%s %8.3f .. %8.3f  (%7.1f us)  %-50s %d" % (qu, (s - a[0]) / 1e6, (e - a[0]) / 1e6, (e - s) / 1e3, n[:50], wgs))
if fast and slow:
    window(fast[len(fast) // 2], "FAST step"); window(slow[len(slow) // 2], "SLOW step")
elif len(sys.argv) > 2 and sys.argv[2] == "window":                # no mode switch in the trace: the window of a middle step
    window(steps[len(steps) // 2][1], "step %d" % (len(steps) // 2))
