#!/usr/bin/env python3
"""The isolated launches of one kernel in a rocprofv3 kernel trace of `bench.py`: runs of at least `minrun` CONSECUTIVE launches
of the same kernel (nothing else started in between -- that is what the roofline leg's timed loops look like; launches inside a
training step never come 20 in a row), with their mean / min / max duration.  The bench line's roofline `achieved` is the
algorithmic work over the mean of its own such loop.  usage: roofline_launches.py <kernel_trace.csv> <kernel substring> [minrun]"""
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
                int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y'])))
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda e: e[0])
sub = sys.argv[2]
minrun = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print("run_first_start_ns,kernel,workgroups,launches,mean_us,min_us,max_us")
i = 0
while i < len(rows):
    if sub not in rows[i][2]:
        i += 1
        continue
    j = i
    while j + 1 < len(rows) and rows[j + 1][2] == rows[i][2] and rows[j + 1][3] == rows[i][3]:
        j += 1
    n = j - i + 1
    if n >= minrun:
        d = [(e[1] - e[0]) / 1e3 for e in rows[i:j + 1]]
        print("%d,%s,%d,%d,%.1f,%.1f,%.1f" % (rows[i][0], rows[i][2], rows[i][3], n, sum(d) / n, min(d), max(d)))
    i = j + 1
