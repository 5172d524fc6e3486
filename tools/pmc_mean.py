#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files (tools only).
usage: pmc_mean.py <dir> [kernel substring ...]"""
import csv, glob, os, sys, collections
root, subs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if subs and not any(s in k for s in subs):
            continue
        g = int(r.get("Grid_Size", 0) or 0) // max(1, int(r.get("Workgroup_Size", 1) or 1))
        acc[(k[:70], g)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), c in sorted(acc.items()):
    print("%s  [%d workgroups]" % (k, g))
    for name, v in sorted(c.items()):
        print("    %-28s mean %16.1f  (n=%d)" % (name, sum(v) / len(v), len(v)))
    if "SQ_WAVE_CYCLES" in c:
        wc = sum(c["SQ_WAVE_CYCLES"]) / len(c["SQ_WAVE_CYCLES"])
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if name in c:
                print("    %-28s %.1f %% of SQ_WAVE_CYCLES" % (name, 100 * sum(c[name]) / len(c[name]) / wc))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        print("    MFMA busy / SQ busy cycles    %.1f %%" % (100 * sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"]) /
                                                      (sum(c["SQ_BUSY_CYCLES"]) / len(c["SQ_BUSY_CYCLES"]))))
