#!/usr/bin/env python3
"""Group the fused kernel's PMC rows of a `rocprofv3 --pmc ... -- python3 tools/placement_exp.py` run by probe (11 dispatches each:
arena 0, then arena i / arena 0 again for i = 1..) and print the per-probe mean of every counter.
usage: placement_pmc.py <rocprof output dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "tile_pass" in r["Kernel_Name"]:
            rows.append(r)
by = defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = sorted({c for d in by.values() for c in d})
print("probe  " + "  ".join(f"{c:>34s}" for c in names))
for p in range(0, len(ids), 11):
    grp = ids[p:p + 11]
    label = "arena 0" if p == 0 else (f"arena {(p // 11 + 1) // 2}" if (p // 11) % 2 == 1 else "arena 0'")
    print(f"{label:8s}" + "  ".join(f"{sum(by[i].get(c, 0.0) for i in grp) / len(grp):34.6g}" for c in names))
