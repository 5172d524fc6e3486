#!/usr/bin/env python3
"""Merge rocprofv3 outputs of tools/hbm_kernels.py into one markdown table.
usage: pmc_table.py <dir with fetch/ write/ trace/ sub-directories> [timings.json]
FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"); WRITE_SIZE is
taken as is; both counters are in KiB.  Rows are per (kernel, grid size): mean over the launches of that shape."""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]


def load(sub, pattern):
    out = []
    for f in glob.glob(os.path.join(root, sub, "**", pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def key(r):
    g = r.get("Grid_Size") or "%sx%sx%s" % (r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    return short(r["Kernel_Name"]), str(g)


cnt = {}
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = defaultdict(list)
    for r in load(sub, "*counter_collection.csv"):
        if r.get("Counter_Name") == cname:
            acc[key(r)].append(float(r["Counter_Value"]))
    cnt[cname] = {k: sum(v) / len(v) for k, v in acc.items()}
dur = defaultdict(list)
for r in load("trace", "*kernel_trace.csv"):
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    dur[(short(r["Kernel_Name"]), str(g))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | grid (threads) | launches | avg us | FETCH_SIZE raw KiB | WRITE_SIZE KiB | HBM-side bytes (2 x fetch + write) | TB/s of HBM traffic |")
print("|---|---|---|---|---|---|---|---|")
keys = sorted(set(cnt["FETCH_SIZE"]) | set(cnt["WRITE_SIZE"]), key=lambda k: -(cnt["FETCH_SIZE"].get(k, 0) + cnt["WRITE_SIZE"].get(k, 0)))
for k in keys:
    f, w = cnt["FETCH_SIZE"].get(k, 0.0), cnt["WRITE_SIZE"].get(k, 0.0)
    d = dur.get(k)
    us = sum(d) / len(d) if d else float("nan")
    total = (2 * f + w) * 1024
    if total < 1e5:
        continue
    print("| `%s` | %s | %s | %.1f | %.0f | %.0f | %.2f MB | %.2f |" % (k[0], k[1], len(d) if d else "-", us, f, w, total / 1e6,
                                                                 total / us / 1e6 if d else float("nan")))
