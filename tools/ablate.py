#!/usr/bin/env python3
"""Ablation timings of the tile pass on one GPU (device-resident buffers, hipEvent timing through the C-ABI).
usage: python tools/ablate.py [pairs]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
L, LB, S = 150, 17, 96
dev = torch.device("cuda", 0)
table = synth.make_sheet(S, 8, dual=True, seed=4)
LIBS = [("cur", None)] + [(os.path.basename(p)[:-3], p) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = []
for name, path in LIBS:
    c = seqkit_amd.Context(0, lib_path=path)
    c.set_barcodes(table, 1)
    ctxs.append((name, c))
ctx = ctxs[0][1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

seq, qual, bc = bench.gen_shard(torch, dev, n, table, seed=1, chunk=2_000_000)
out = [torch.empty_like(seq[0]) for _ in range(2)]
lk = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]
assign = torch.empty((n,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def mate(i, mask, trim):
    return {"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0,
            "out_seq": out[i].data_ptr() if mask else 0, "lowest_k": lk[i].data_ptr() if trim else 0}


ONLY = os.environ.get("ABLATE_ONLY", "")


def timeit(name, fn, nbytes, iters=5, rounds=5):
    if ONLY and not any(tok in name for tok in ONLY.split(",")):
        return
    for _ in range(2):
        fn()
    ctx.sync()
    ts = []
    for _ in range(rounds):
        ctx.timer_start()
        for _ in range(iters):
            fn()
        ts.append(ctx.timer_stop() / iters)
    ts.sort()
    ms, best = ts[len(ts) // 2], ts[0]
    print(f"{name:36s} med {ms:7.3f} ms {nbytes / ms / 1e6:7.1f} GB/s | min {best:7.3f} ms {nbytes / best / 1e6:7.1f} GB/s | {n / ms / 1e3:8.1f} M clusters/s", flush=True)


ROUNDS = int(os.environ.get("ABLATE_ROUNDS", "2"))
mb = 20
for rnd in range(ROUNDS):          # interleave the libraries so that clock drift hits them alike
    for name, ctx in ctxs:
        print(f"--- {name} (round {rnd})")
        timeit("mask_flat 1 mate (450 B)", lambda: ctx.mask_by_quality_dev(seq[0].data_ptr(), qual[0].data_ptr(), L, n, mb, out[0].data_ptr()), 450 * n)
        timeit("tile: mask only 1 mate (450 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, True, False)]), 450 * n)
        timeit("tile: trim only 1 mate (152 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, False, True)]), 152 * n)
        timeit("tile: mask+trim 1 mate (452 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, True, True)]), 452 * n)
        timeit("tile: mask+trim 2 mates (904 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, True, True), mate(1, True, True)]), 904 * n)
        timeit("tile: demux only (21 B)", lambda: ctx.demux_assign_dev(bc.data_ptr(), LB, n, assign.data_ptr()), 21 * n)
        timeit("tile: full fused 2 mates (925 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, True, True), mate(1, True, True)], bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr()), 925 * n)
        timeit("tile: full fused 1 mate (473 B)", lambda: ctx.fused_pass_dev(n, L, mb, [mate(0, True, True)], bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr()), 473 * n)
