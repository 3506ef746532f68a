#!/usr/bin/env python3
"""Resident waves per CU of the fused pass, compared IN ONE PROCESS on the same buffers (the run-to-run spread between
processes comes from where the allocations land, so shapes must be compared on one placement).
usage: python tools/waves_exp.py [clusters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
seq, qual, bc = bench.gen_shard(torch, dev, n, table, seed=4000, chunk=2_000_000)
out = [torch.empty_like(seq[0]) for _ in range(2)]
lk = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]
assign = torch.empty((n,), dtype=torch.int32, device=dev)
mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr(), "lowest_k": lk[i].data_ptr()} for i in range(2)]
def mate(i, mask, trim):
    return {"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr() if mask else 0, "lowest_k": lk[i].data_ptr() if trim else 0}


CASES = {
    "fused 2 mates (925 B)": (925, lambda: ctx.fused_pass_dev(n, L, 20, mates, bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr())),
    "mask+trim 2 mates (904 B)": (904, lambda: ctx.fused_pass_dev(n, L, 20, mates)),
    "fused 1 mate (473 B)": (473, lambda: ctx.fused_pass_dev(n, L, 20, [mates[0]], bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr())),
    "mask+trim 1 mate (452 B)": (452, lambda: ctx.fused_pass_dev(n, L, 20, [mate(0, True, True)])),
    "trim 1 mate (152 B)": (152, lambda: ctx.fused_pass_dev(n, L, 20, [mate(0, False, True)])),
    "mask 1 mate, tile pass (450 B)": (450, lambda: ctx.fused_pass_dev(n, L, 20, [mate(0, True, False)])),
}
ONLY = os.environ.get("WAVES_ONLY", "fused 2")


def probe(run):
    for _ in range(2):
        run()
    ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(3):
            run()
        ts.append(ctx.timer_stop() / 3)
    return sorted(ts)[1]


shapes = [("0", "0"), ("8", "2"), ("8", "1"), ("4", "2"), ("4", "3"), ("2", "4"), ("0", "0")]
if os.environ.get("WAVES_SHAPES"):
    shapes = [tuple(x.split(":")) for x in os.environ["WAVES_SHAPES"].split(",")]
for name, (nbytes, run) in CASES.items():
    if not any(tok in name for tok in ONLY.split(",")):
        continue
    for rnd in range(2):
        for nw, wg in shapes:
            os.environ["SK_TILE_WAVES"], os.environ["SK_TILE_WGS"] = nw, wg
            ms = probe(run)
            print(f"{name:32s} round {rnd} waves/WG={nw} WGs/CU={wg}: {ms:7.3f} ms  {nbytes * n / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
