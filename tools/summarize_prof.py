#!/usr/bin/env python3
"""Keep the rows of a rocprofv3 --kernel-trace --stats run that concern this repo's kernels.

usage: summarize_prof.py <rocprof output dir> <profiles/out_prefix> [<bench JSON line of the traced run>]
Writes <prefix>_kernel_stats.csv (sk:: kernels + the total of everything else),
<prefix>_kernel_trace_head.csv (first dispatches of each sk:: kernel: grid, LDS, VGPR/SGPR counts) and, with the traced
run's own JSON line, <prefix>_timed_steps.txt: the trace's durations of that run's timed steps (its last `steps`
dispatches of the headline kernel) next to the kernel time the run measured itself with HIP events.
"""
import csv
import glob
import os
import sys


def timed_steps(trace_csv, prefix, bench_json):
    import json
    try:
        line = json.load(open(bench_json))
    except Exception:
        return
    kern, steps = line["roofline"]["kernel"], int(line["steps"])
    rows = [r for r in csv.DictReader(open(trace_csv)) if kern in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-steps:]]
    allv = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    with open(prefix + "_timed_steps.txt", "w") as f:
        f.write(f"kernel {kern}: {len(rows)} dispatches in the traced run (placement probes + warm-up + {steps} timed steps)\n")
        f.write(f"  all dispatches            : mean {sum(allv) / len(allv) / 1e6:.4f} ms  min {min(allv) / 1e6:.4f}  max {max(allv) / 1e6:.4f}\n")
        f.write(f"  the {steps} timed steps (trace) : mean {sum(last) / len(last) / 1e6:.4f} ms  min {min(last) / 1e6:.4f}  max {max(last) / 1e6:.4f}\n")
        f.write(f"  same run, HIP events      : kernel_ms {line['roofline']['kernel_ms']}  ms_per_step {line['ms_per_step']}  frac {line['roofline']['frac']}\n")
        f.write(f"  placement of that run     : { {k: v for k, v in (line['config'].get('placement') or {}).items() if k != 'what'} }\n")


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
        if len(sys.argv) > 3:
            timed_steps(trace[0], prefix, sys.argv[3])


if __name__ == "__main__":
    main()
