#!/usr/bin/env python3
"""Backward-pass window of one step from a rocprofv3 kernel trace: per queue, what ran between the last ROIAlign adjoint
and the optimiser (launch count, busy time, the longest kernels).  usage: bwd_window.py <kernel_trace.csv> [step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
             int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X']), r['Queue_Id']) for r in rows)
sgd = [e for e in ev if e[2].startswith('sgd_kernel')]
t0, t1 = sgd[step][1], sgd[step + 1][1]
seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("step wall %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(seg)))
for q in sorted(set(e[4] for e in seg)):
    qs = [e for e in seg if e[4] == q]
    print("queue %s: %d launches, busy %.3f ms, first start +%.3f ms, last end +%.3f ms" % (
        q, len(qs), sum(e[1] - e[0] for e in qs) / 1e6, (qs[0][0] - t0) / 1e6, (qs[-1][1] - t0) / 1e6))
big = [e for e in seg if e[1] - e[0] > 300e3]
print("kernels longer than 0.3 ms, in start order:")
for e in big:
    print("  q%s +%8.3f .. +%8.3f ms  %7.3f ms  %-40s wgs=%d" % (e[4], (e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2][:40], e[3]))
r = [e for e in seg if e[2].startswith('roialign_kernel<true>')]
if r:
    r = r[-1]
    print("last ROIAlign adjoint ends at +%.3f ms" % ((r[1] - t0) / 1e6))
    bw = [e for e in seg if e[0] >= r[1] and e[1] - e[0] <= 300e3]
    g = collections.defaultdict(lambda: [0, 0])
    for e in bw:
        g[(e[4], e[2], e[3])][0] += 1; g[(e[4], e[2], e[3])][1] += e[1] - e[0]
    print("small kernels after it:")
    for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:12]:
        print("  q%s %-44s wgs=%6d n=%4d sum=%7.3f ms mean=%6.1f us" % (k[0], k[1][:44], k[2], c, t / 1e6, t / c / 1e3))
