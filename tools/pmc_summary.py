#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter rows of the sk:: kernels.  usage: pmc_summary.py <dir> [<dir> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sk::" in k:
                acc[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
