#!/usr/bin/env python3
"""Keep the rows of a rocprofv3 --kernel-trace --stats run that concern this repo's kernels.

usage: summarize_prof.py <rocprof output dir> <profiles/out_prefix>
Writes <prefix>_kernel_stats.csv (sk:: kernels + the total of everything else) and
<prefix>_kernel_trace_head.csv (first dispatches of each sk:: kernel: grid, LDS, VGPR/SGPR counts).
"""
import csv
import glob
import os
import sys


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        ours = [r for r in rows if ("sk::" in r["Name"])]
        other_ns = sum(int(r["TotalDurationNs"]) for r in rows if not ("sk::" in r["Name"]))
        with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in ours:
                w.writerow(r)
            w.writerow({"Name": "(all other kernels: torch data generation, copies, fills)", "Calls": sum(int(r["Calls"]) for r in rows if not ("sk::" in r["Name"])),
                        "TotalDurationNs": other_ns})
    if trace:
        seen = {}
        with open(prefix + "_kernel_trace_head.csv", "w", newline="") as f:
            rd = csv.DictReader(open(trace[0]))
            w = csv.DictWriter(f, fieldnames=rd.fieldnames + ["Duration_Ns"])
            w.writeheader()
            for r in rd:
                k = r["Kernel_Name"]
                if "sk::" not in k:
                    continue
                seen[k] = seen.get(k, 0) + 1
                if seen[k] <= 8:
                    r["Duration_Ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    w.writerow(r)


if __name__ == "__main__":
    main()
