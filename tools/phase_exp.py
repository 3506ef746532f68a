#!/usr/bin/env python3
"""Does the fused pass care about the PHASE between the two mates' streams?  Mate 2's arrays are handed over shifted by k
rows (the kernel then reads mate 2 of cluster r + k next to mate 1 of cluster r): same work, same bytes, one process, one
placement.  usage: python tools/phase_exp.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = 62_500_000
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
seq, qual, bc = bench.gen_shard(torch, dev, n, table, seed=4000, chunk=2_000_000)
out = [torch.empty_like(seq[0]) for _ in range(2)]
lk = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]
assign = torch.empty((n,), dtype=torch.int32, device=dev)
KMAX = 8_000_000
m = n - KMAX


def run(k):
    k = k // 64 * 64
    mates = [{"seq": seq[0].data_ptr(), "qual": qual[0].data_ptr(), "len": 0, "out_seq": out[0].data_ptr(), "lowest_k": lk[0].data_ptr()},
             {"seq": seq[1].data_ptr() + k * L, "qual": qual[1].data_ptr() + k * L, "len": 0, "out_seq": out[1].data_ptr() + k * L, "lowest_k": lk[1].data_ptr() + k * 2}]
    ctx.fused_pass_dev(m, L, 20, mates, bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr())


def probe(k):
    for _ in range(2):
        run(k)
    ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(3):
            run(k)
        ts.append(ctx.timer_stop() / 3)
    return sorted(ts)[1]


for rnd in range(2):
    for k in (0, 64, 640, 6400, 64000, 640_000, 1_000_000, 2_000_000, 3_500_000, 5_000_000, 7_999_936, 0):
        ms = probe(k)
        print(f"round {rnd} shift {k:9d} rows ({k * L / 1e6:8.1f} MB): {ms:7.3f} ms  {925 * m / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
