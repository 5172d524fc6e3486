#!/usr/bin/env python3
"""Gaps on the main hardware queue of one training step (rocprofv3 kernel trace CSV): every idle interval longer than
`min_us` with the kernel before and after it, and what the other queues ran meanwhile.  Profile the TAPED step
(MRCNN_TRAIN_TAPE=1 tools/mode_timing.py ...): issued eagerly under the profiler, the host falls behind and the gaps are its.
usage: queue_gaps.py <kernel_trace.csv> [step] [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
             int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // int(r['Workgroup_Size_X']), r['Queue_Id']) for r in rows)
sgd = [e for e in ev if e[2].startswith('sgd_kernel')]
t0, t1 = sgd[step][1], sgd[step + 1][1]
seg = [e for e in ev if e[0] >= t0 and e[1] <= t1]
qs = {}
for e in seg:
    qs.setdefault(e[4], []).append(e)
main = max(qs, key=lambda q: sum(e[1] - e[0] for e in qs[q]))
print("step wall %.3f ms; main queue %s busy %.3f ms in %d launches" % ((t1 - t0) / 1e6, main, sum(e[1] - e[0] for e in qs[main]) / 1e6, len(qs[main])))
prev_end, prev = t0, None
total = 0.0
for e in qs[main]:
    gap = (e[0] - prev_end) / 1e3
    if gap > 0:
        total += gap
    if gap >= min_us:
        others = [o for o in seg if o[4] != main and o[0] < e[0] and o[1] > prev_end]
        busy = sum(min(o[1], e[0]) - max(o[0], prev_end) for o in others) / 1e3
        names = {}
        for o in others:
            names[o[2][:36]] = names.get(o[2][:36], 0) + 1
        print("+%8.3f ms  gap %7.1f us  after %-34s before %-34s | other queues %6.1f us busy: %s" % (
            (prev_end - t0) / 1e6, gap, prev[2][:34] if prev else "-", e[2][:34], busy,
            ", ".join("%s x%d" % kv for kv in sorted(names.items(), key=lambda kv: -kv[1])[:3])))
    if e[1] > prev_end:
        prev_end, prev = e[1], e
print("sum of all main-queue gaps: %.3f ms" % (total / 1e3))
