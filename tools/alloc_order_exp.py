#!/usr/bin/env python3
"""Does it matter WHEN the blocked buffers are allocated?  first: the two big buffers are the process's first device
allocations (then filled chunk by chunk); after: they are allocated while the 38 GB of SoA matrices they are packed from
are resident (what bench.py did).  One process per mode, so that each starts from the same memory state.
usage: python tools/alloc_order_exp.py first|after [clusters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 62_500_000
L, LB = 150, 17
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
nt = (n + 63) // 64
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
if mode == "first":
    bin_ = torch.zeros(nt * lay.in_block, dtype=torch.uint8, device=dev)
    bout = torch.empty(nt * lay.out_block, dtype=torch.uint8, device=dev)
    ctx = seqkit_amd.Context(0)
    ctx.set_barcodes(table, 1)
    step_t = 31250                                     # tiles per chunk = 2 M clusters
    for t0 in range(0, nt, step_t):
        tn = min(step_t, nt - t0)
        seq, qual, bc = bench.gen_shard(torch, dev, tn * 64, table, seed=4000 + t0, chunk=tn * 64)
        part, _ = bench.pack_blocked(torch, lay, seq, qual, bc, tn)
        bin_[t0 * lay.in_block:(t0 + tn) * lay.in_block] = part
        del seq, qual, bc, part
else:
    ctx = seqkit_amd.Context(0)
    ctx.set_barcodes(table, 1)
    seq, qual, bc = bench.gen_shard(torch, dev, nt * 64, table, seed=4000, chunk=2_000_000)
    bin_, bout = bench.pack_blocked(torch, lay, seq, qual, bc, nt)
    del seq, qual, bc
torch.cuda.empty_cache()
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
torch.cuda.synchronize()


def run():
    ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())


for _ in range(3):
    run()
ctx.sync()
ts = []
for _ in range(5):
    ctx.timer_start()
    for _ in range(3):
        run()
    ts.append(ctx.timer_stop() / 3)
ms = sorted(ts)[2]
print(f"{mode:6s}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}% of 8 TB/s   in @ {bin_.data_ptr():#x} out @ {bout.data_ptr():#x}", flush=True)
