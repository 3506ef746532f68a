#!/usr/bin/env python3
"""The numbers DESIGN.md §6 quotes, from the round's profile set: prints the extra.rates table (markdown) and the headline /
census figures.  usage: python tools/design_numbers.py [r04] [--write]   (--write replaces the table between DESIGN.md's markers)"""
import json
import os
import re
import sys

tag = next((a for a in sys.argv[1:] if not a.startswith("-")), "r06")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(root, "profiles", f"{tag}_bench.json")).read().strip().splitlines()[-1])
bound = {"cfg1": "HBM", "cfg2 worst": "scan issue, then HBM", "cfg2: trim": "25 µs of streaming + 6 µs of launch and events per call", "cfg2 read": "HBM", "fused single-end:": "HBM",
         "fused single-end, ragged": "HBM", "fused paired": "HBM", "cfg3: demul": "memory side as placed; 20 µs of streaming + 6 µs of launch and events", "cfg3 with": "the same",
         "cfg3 sheet": "memory side as placed (§3.2b)", "demultiplex only 10M x 17ch, 96 dual-index, a mixed": "as the 384-sample sheet: three lookups per read",
         "demultiplex only 10M x 17ch, 96": "memory side as placed + the call's fixed part", "96 dual-index with": "the same", "96 dual-index, 100M": "memory side as placed (§3.2b)",
         "demultiplex only 10M x 17ch, 384": "three lookups per read, then the memory side", "cfg5": "HBM", "f2:": "HBM", "f4:": "HBM as placed (LDS-tile kernel, §3.8)",
         "f3: census 32M rows (1 M drawn rows x 32: rounds 3-5's input), clean": "front kernel (LDS pipe + issue) 0.18 ms + combine 0.05 ms",
         "f3: census 32M rows (1 M drawn rows x 32: rounds 3-5's input), noisy": "front kernel 0.19–0.21 ms + combine 0.11 ms",
         "f3: census 32M independently": "the distinct keys' inserts into the HBM table: ≈ 25 G scattered agent-scope operations/s chip-wide (§8)",
         "f3: census 32M rows, every": "memory side: one CAS + two stores per new key"}
rows = []
for x in d["extra"]["rates"]:
    b = next((v for k, v in bound.items() if x["config"].startswith(k)), "")
    placed = f" (as placed {x['frac_as_placed']:.3f})" if x.get("frac_as_placed") else ""
    if "frac_warm" in x:                                       # rows timed from HBM: buffer sets in rotation, one event pair per call
        placed += f" from HBM; pipelined {x['frac_pipelined']:.3f}, on-die replay {x['frac_warm']:.3f} ({x['ms_warm'] * 1000:.1f} µs)"
    if "frac_many" in x:                                       # the buffer sets as the batches of ONE many-batch call
        placed += f"; per batch of one many-batch call {x['frac_many']:.3f} ({x['ms_many'] * 1000:.1f} µs)" + (f", three such calls pipelined {x['frac_many_pipelined']:.3f}" if "frac_many_pipelined" in x else "")
    if "reset_ms" in x:                                        # census: sk_census_reset before the count
        placed += f"; reset {x['reset_ms'] * 1000:.0f} µs, reset + count {x['reset_plus_count_ms'] * 1000:.0f} µs"
    rows.append(f"| {x['config']} | {x['ms'] * 1000:.1f} µs, {x['G_units_per_s']:.1f} G units/s | {x['frac']:.3f}{placed} | {b} |")
table = "| config (device-resident; `bench.py` `extra.rates`) | rate | of 8 TB/s | bound |\n|---|---|---|---|\n" + "\n".join(rows)
r = d["roofline"]
print(f"headline: value {d['value']} M reads/s, ms_per_step {d['ms_per_step']}, kernel_ms {r['kernel_ms']}, achieved {r['achieved']} GB/s, frac {r['frac']}, "
      f"frac_as_placed {r['frac_as_placed']}, read_frac {r['read_frac']}, traffic {r['traffic']}")
print("placement:", d["config"]["placement"].get("ms_before"), "->", d["config"]["placement"].get("ms_after"))
print("cpu all cores:", d["cpu_baseline"].get("all_cores"))
print("cpu:", d["cpu_baseline"]["value"], {k: v["M_reads_per_s"] for k, v in d["cpu_baseline"].get("faithful", {}).items() if isinstance(v, dict)})
print(table)
for x in d["extra"].get("bam_files", []):
    print(f"bam file: {x['ms']:.1f} ms, {x['M_records_per_s']:.1f} M records/s, {x['compressed_GBps']:.2f} GB/s compressed, {x['inflated_GBps']:.1f} GB/s inflated; {x['config']}")
if "--write" in sys.argv:
    p = os.path.join(root, "DESIGN.md")
    s = open(p).read()
    s2 = re.sub(r"(<!-- rates:begin -->\n).*?(\n<!-- rates:end -->)", lambda m: m.group(1) + table + m.group(2), s, flags=re.S)
    assert s2 != s or table in s, "markers not found"
    open(p, "w").write(s2)

# ---- the facts DESIGN.md §6's paragraphs quote, from the same profile set (printed, not written: the paragraphs are prose)
import csv
import glob


def kstat(name, pick=None):
    rows = list(csv.reader(open(os.path.join(root, "profiles", f"{tag}_{name}_rocprof_kernel_stats.csv"))))[1:]
    for r in rows:
        if r[1] and r[3] and (pick is None or pick in r[0]):
            return int(r[1]), float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3
    return None


def pmc(name, kernel=None):
    out, cur = {}, None
    for line in open(os.path.join(root, "profiles", f"{tag}_{name}_pmc_summary.txt")):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        if kernel is None or kernel in (cur or ""):
            k, _, rest = line.strip().partition(" ")
            out.setdefault(k, float(rest.split("mean=")[1]))
    return out


print("\n== facts for §6")
print(open(os.path.join(root, "profiles", f"{tag}_rocprof_timed_steps.txt")).read().strip())
for name, gb, label in (("lut_cfg3_100m", 1.2, "cfg3 sheet 100 M"), ("lut_dual_100m", 2.1, "96 dual-index 100 M"), ("k_mask", 7.2, "mask_flat"), ("k_bam", 2.8, "bam_flag_tlen"),
                        ("k_fragments", 2.825, "bam_fragments"), ("k_sequence152", 6.144, "sequence pitch 152"), ("k_sequence148", 6.007, "sequence pitch 148")):
    n, mean, lo, hi = kstat(name)
    p = pmc(name)
    traffic = (p.get("FETCH_SIZE", 0) * 2048 + p.get("WRITE_SIZE", 0) * 1024) / 1e9
    wait = p.get("SQ_WAIT_ANY", 0) / p["SQ_WAVE_CYCLES"] if p.get("SQ_WAVE_CYCLES") else float("nan")
    print(f"{label}: {mean:.1f} us mean of {n} ({lo:.1f}-{hi:.1f}) = {gb / mean * 1e6 / 8000:.3f} of 8 TB/s; traffic {traffic:.2f} GB = {traffic / gb:.2f} x; waves wait {wait:.2f}")
for name in ("k_inflate_random", "k_inflate_sorted", "k_deflate"):
    for pick in ("inflate_kernel", "crc", "deflate_kernel"):
        r = kstat(name, pick)
        if r:
            print(f"{name} {pick}: {r[1] / 1e3:.2f} ms mean of {r[0]}")
    p = pmc(name)
    print(f"   traffic of the first kernel: {p.get('FETCH_SIZE', 0) * 2048 / 1e9:.2f} GB read + {p.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB written")
p = {}
cur = None
for line in open(os.path.join(root, "profiles", f"{tag}_census_indep_pmc_noisy_indep.txt")):
    if not line.startswith(" "):
        cur = line.strip()
    elif "FETCH_SIZE" in line or "WRITE_SIZE" in line or "LDS_BANK" in line or "ACTIVE_INST_LDS" in line:
        p[(cur, line.split()[0])] = float(line.split("mean=")[1])
tot = 0.0
for k in sorted({c for c, _ in p}):
    rd, wr = p.get((k, "FETCH_SIZE"), 0) * 2048 / 1e6, p.get((k, "WRITE_SIZE"), 0) * 1024 / 1e6
    tot += rd + wr
    print(f"census indep noisy {k}: {rd:.0f} + {wr:.0f} MB; bank conflict cycles {p.get((k, 'SQ_LDS_BANK_CONFLICT'), 0) / 1e6:.1f} M, LDS instructions {p.get((k, 'SQ_ACTIVE_INST_LDS'), 0) / 1e6:.1f} M")
print(f"   total {tot / 1e3:.2f} GB = {tot / 544:.2f} x of 544 MB")
print("".join(open(os.path.join(root, "profiles", f"{tag}_census_indep_trace_noisy_indep.txt")).readlines()[-6:]).rstrip())
for f in ("lut_repro.txt", "many_rate.txt", "inflate_rate.txt", "deflate_rate.txt"):
    print(f"-- {f}")
    print("".join(open(os.path.join(root, "profiles", f"{tag}_{f}")).readlines()[-6:]).rstrip()[:1500])
for f, pat in (("bam_gpu.txt", ("wall", "handled", "waited")), ("deflate_e2e.txt", ("hip  ", "cpu user", "==")), ("demux_prof.txt", ("==", "hip  ", "demultiplex:", "process:")), ("inflate_stamps.txt", ("group loop", "flushed bytes", "outside")),
               ("gpu_tests.txt", ("passed",))):
    print(f"-- {f}")
    for line in open(os.path.join(root, "profiles", f"{tag}_{f}")):
        if any(x in line for x in pat):
            print("  " + line.rstrip()[:230])
