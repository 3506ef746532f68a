#!/usr/bin/env python3
"""sk_bam_file_reduce on one file, a few times: the stages' times and rates.  usage: bam_file_info.py <file.bam>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import seqkit_amd  # noqa: E402

ctx = seqkit_amd.Context(0)
for _ in range(int(os.environ.get("BAM_INFO_REPS", "3"))):
    t0 = time.perf_counter()
    handled, counters, hist, total, info = ctx.bam_file_reduce(sys.argv[1], 5000)
    dt = time.perf_counter() - t0
    print(f"handled {handled}: {dt * 1e3:.1f} ms wall; {info[0] / 1e9:.2f} GB compressed -> {info[1] / 1e9:.2f} GB in {int(info[2])} blocks, {int(info[3])} records, "
          f"{int(info[4])} blocks by zlib, {int(info[5])} walk rounds; read+copy {info[6]:.1f} ms, device tail {info[7]:.1f} ms; "
          f"{info[3] / dt / 1e6:.1f} M records/s, {info[0] / dt / 1e9:.2f} GB/s compressed; counters {counters.tolist()} hist total {total}")
