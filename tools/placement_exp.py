#!/usr/bin/env python3
"""Same bytes, same virtual layout, different allocations: does the fused-pass time follow the (physical) placement?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(os.environ.get("N", "62500000"))
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
sizes = [n * L] * 4 + [n * LB] + [n * L] * 2 + [n * 2] * 2 + [n * 4]
step = [(s + 4095) // 4096 * 4096 for s in sizes]
offs = [sum(step[:i]) for i in range(len(step))]
total = sum(step) + 4096


def views(arena):
    cut = [arena[o:o + s] for o, s in zip(offs, sizes)]
    return cut


def probe(arena):
    p = [arena.data_ptr() + o for o in offs]
    mates = [{"seq": p[0], "qual": p[1], "len": 0, "out_seq": p[5], "lowest_k": p[7]},
             {"seq": p[2], "qual": p[3], "len": 0, "out_seq": p[6], "lowest_k": p[8]}]
    run = lambda: ctx.fused_pass_dev(n, L, 20, mates, bc=p[4], bc_stride=LB, assign=p[9])
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


a = torch.empty(total, dtype=torch.uint8, device=dev)
c = views(a)
bench.gen_shard(torch, dev, n, table, seed=4000, chunk=2_000_000,
                into=([c[0].view(n, L), c[2].view(n, L)], [c[1].view(n, L), c[3].view(n, L)], c[4].view(n, LB)))
torch.cuda.synchronize()
print(f"arena 0 {a.data_ptr():#x}: {probe(a):.3f} ms", flush=True)
keep = []
for i in range(1, int(os.environ.get("ARENAS", "4"))):
    b = torch.empty(total, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    print(f"arena {i} {b.data_ptr():#x}: {probe(b):.3f} ms   (arena 0 again: {probe(a):.3f} ms)", flush=True)
    keep.append(b)          # keep them all alive so that every arena is a different physical range
