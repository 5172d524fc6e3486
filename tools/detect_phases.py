#!/usr/bin/env python3
"""Kernel time of eager detect passes (batch 1) per kernel, from a rocprofv3 kernel trace of tools/profile_detect.py <n>.
usage: detect_phases.py <kernel_trace.csv> <n passes> (tools only)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
g = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    g[k][0] += 1; g[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(t for c, t in g.values())
print("%d launches and %.3f ms of kernel time per pass" % (len(rows) // n, tot / n / 1e6))
for k, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-62s %5.1f launches  %.3f ms" % (k, c / n, t / n / 1e6))
