#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 rocpd database (kernel trace): union busy time, idle gaps,
and the kernels in start order with the number of workgroups (tools only)."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = c.execute("select start, end, name, queue_id, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) from kernels order by start").fetchall()
sgd = [i for i, r in enumerate(rows) if r[2].startswith('sgd_kernel')]
# last sgd launch of step k .. last of step k+1: find groups of consecutive sgd kernels
ends = [sgd[i] for i in range(len(sgd)) if i + 1 == len(sgd) or sgd[i + 1] != sgd[i] + 1 and rows[sgd[i + 1]][0] - rows[sgd[i]][1] > 2e6]
a, b = ends[step], ends[step + 1]
seg = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
print("step wall %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
busy = 0; cur_e = t0; gaps = []
for s, e, n, q, wg in seg:
    if s > cur_e:
        gaps.append((s - cur_e, n)); cur_e = s
    if e > cur_e:
        busy += e - cur_e; cur_e = e
print("busy (any queue) %.3f ms, idle %.3f ms in %d gaps" % (busy / 1e6, sum(g for g, _ in gaps) / 1e6, len(gaps)))
byq = collections.Counter()
for s, e, n, q, wg in seg: byq[q] += e - s
print("summed kernel time per queue:", {q: round(v / 1e6, 3) for q, v in byq.items()})
if len(sys.argv) > 3:
    for s, e, n, q, wg in seg:
        print("%9.3f %8.1f us  q%d  wg=%-6d %s" % ((s - t0) / 1e6, (e - s) / 1e3, q, wg, n[:70]))
else:
    acc = collections.Counter(); cnt = collections.Counter()
    for s, e, n, q, wg in seg: acc[n[:60]] += e - s; cnt[n[:60]] += 1
    for n, v in acc.most_common(30):
        print("%-60s n=%4d  %8.3f ms  avg %6.1f us" % (n, cnt[n], v / 1e6, v / cnt[n] / 1e3))
