#!/usr/bin/env python3
"""Which part of the tile-blocked pass costs bandwidth: the same shard with the trim scan and / or the barcode phase switched
off (the layout flags), each against its own algorithmic bytes, next to the flat mask kernel on the same unit.
usage: python tools/blk_parts.py [clusters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
nt = (n + 63) // 64
seq, qual, bc = bench.gen_shard(torch, dev, nt * 64, table, seed=4000, chunk=2_000_000)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)


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


cases = [("mask + trim + demultiplex", capi.SK_BLK_MASK | capi.SK_BLK_TRIM, LB, 925), ("mask + trim", capi.SK_BLK_MASK | capi.SK_BLK_TRIM, 0, 904),
         ("mask + demultiplex", capi.SK_BLK_MASK, LB, 921), ("mask only (blocked pass)", capi.SK_BLK_MASK, 0, 900),
         ("trim + demultiplex", capi.SK_BLK_TRIM, LB, 325)]
for rnd in range(2):
    for name, flags, bstride, nbytes in cases:
        lay = capi.blocked_layout(2, L, bstride, flags)
        bin_, bout = bench.pack_blocked(torch, lay, seq, qual, bc, nt)
        torch.cuda.synchronize()
        ms = probe(lambda: ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr()))
        print(f"round {rnd} {name:28s}: {ms:7.3f} ms  {nbytes} B/cluster  {nbytes * n / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
        del bin_, bout
        torch.cuda.empty_cache()
    out = torch.empty_like(seq[0])
    ms = probe(lambda: (ctx.mask_by_quality_dev(seq[0].data_ptr(), qual[0].data_ptr(), L, n, 20, out.data_ptr()),
                        ctx.mask_by_quality_dev(seq[1].data_ptr(), qual[1].data_ptr(), L, n, 20, out.data_ptr())))
    print(f"round {rnd} {'mask_flat, both mates (2 launches)':28s}: {ms:7.3f} ms  900 B/cluster  {900 * n / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
    del out
