#!/usr/bin/env python3
"""The last kernels of a rocprofv3 --kernel-trace run whose names contain one of the given words.  usage: trace_tail.py <dir> <word> [<word> ...]"""
import csv
import glob
import os
import sys

t = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
words = [w.lower() for w in sys.argv[2:]]
rows = [r for r in csv.DictReader(open(t)) if any(w in r["Kernel_Name"].lower() for w in words)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows[-int(os.environ.get("TAIL", "14")):]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sk::", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:12.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {name}")
