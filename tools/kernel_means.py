#!/usr/bin/env python3
"""Mean duration per (kernel, workgroups) of a rocprofv3 kernel trace, optionally only kernels whose name contains a
substring (tools only).  usage: kernel_means.py <kernel_trace.csv> [substring ...]"""
import csv, sys, collections
g = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    wgs = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']))
    g[(n[:56], wgs)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, w), v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    print("%-58s %8d %6d %10.1f %10.3f" % (n, w, len(v), sum(v) / len(v), sum(v) / 1e3))
