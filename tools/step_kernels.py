#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace, grouped by (kernel, grid): launches, summed and mean duration; and the
phase boundaries (trunk+RPN forward = step start .. ProposalLayer; trunk backward = last ROIAlign adjoint .. optimiser).
usage: step_kernels.py <kernel_trace.csv> [step index]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
             int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']), int(r['Workgroup_Size_X'])) for r in rows)
sgd = [e for e in ev if e[2].startswith('sgd_kernel')]
t0, t1 = sgd[step][1], sgd[step + 1][1]
seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("step wall %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(seg)))
def first(name): return next((e for e in seg if e[2].startswith(name)), None)
def last(name): return next((e for e in reversed(seg) if e[2].startswith(name)), None)
p = first('select_sort_decode') or first('topk')
if p: print("trunk + RPN forward (start .. ProposalLayer): %.3f ms, %d launches" % ((p[0] - seg[0][0]) / 1e6, sum(1 for e in seg if e[1] <= p[0])))
r = last('roialign_kernel<true>') or last('roialign_bwd_gather')
if r: print("trunk backward (last ROIAlign adjoint .. optimiser): %.3f ms, %d launches" % ((t1 - r[1]) / 1e6, sum(1 for e in seg if e[0] >= r[1])))
# busy time (union of intervals) and idle
iv = sorted((e[0], e[1]) for e in seg); busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print("GPU busy (union of kernels) %.3f ms, idle %.3f ms" % (busy / 1e6, (t1 - t0 - busy) / 1e6))
g = collections.defaultdict(lambda: [0, 0])
for s, e, n, grid, wg in seg:
    g[(n, grid // wg)][0] += 1; g[(n, grid // wg)][1] += e - s
print("%-52s %9s %6s %10s %9s" % ("kernel", "workgroups", "n", "sum ms", "mean us"))
for (n, wgs), (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%-52s %9d %6d %10.3f %9.1f" % (n[:52], wgs, c, t / 1e6, t / c / 1e3))
