#!/usr/bin/env python3
"""Does the fused-pass time depend on where the arrays sit in memory?  Re-allocate with varying offsets in ONE process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(os.environ.get('N', '8000000'))
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
g = torch.Generator(device=dev)
g.manual_seed(1)


def alloc(pad_bytes):
    """one big arena; arrays carved at offsets separated by pad_bytes extra"""
    sz = n * L
    sizes = [sz, sz, sz, sz, sz, sz, n * LB, n * 2, n * 2, n * 4]
    total = sum((s + pad_bytes + GRAN - 1) // GRAN * GRAN for s in sizes) + 4096
    arena = torch.empty(total, dtype=torch.uint8, device=dev)
    ptrs, off = [], 0
    for s in sizes:
        ptrs.append(arena.data_ptr() + off)
        off += (s + pad_bytes + GRAN - 1) // GRAN * GRAN
    return arena, ptrs


PADS = [int(x) for x in os.environ.get('PADS', '').split(',') if x] or [0, 4096, 65536, 1 << 20, 3 << 20, (1 << 21) + 4096 * 37, 1 << 24, 12345 * 4096]
GRAN = int(os.environ.get('GRAN', '4096'))
for pad in PADS:
    arena, p = alloc(pad)
    # fill inputs with plausible bytes
    arena.random_(33, 74, generator=g)
    torch.cuda.synchronize()
    mates = [{"seq": p[0], "qual": p[1], "len": 0, "out_seq": p[4], "lowest_k": p[7]},
             {"seq": p[2], "qual": p[3], "len": 0, "out_seq": p[5], "lowest_k": p[8]}]

    def run():
        ctx.fused_pass_dev(n, L, 20, mates, bc=p[6], bc_stride=LB, assign=p[9])

    for _ in range(3):
        run()
    ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(4):
            run()
        ts.append(ctx.timer_stop() / 4)
    ts.sort()
    print(f"pad {pad:>10d}  arena {arena.data_ptr():#x}  med {ts[1]:.3f} ms  min {ts[0]:.3f}  {925 * n / ts[1] / 1e6:.0f} GB/s", flush=True)
    del arena
    torch.cuda.empty_cache()
