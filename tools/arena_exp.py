#!/usr/bin/env python3
"""Everything a pass touches carved from ONE device allocation: the row-major matrices (nine arrays) and the tile-blocked
pair, timed in one process next to separately allocated buffers of the same content.
usage: python tools/arena_exp.py [clusters]"""
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
table = synth.make_sheet(96, 8, dual=True, seed=4)
nt = (n + 63) // 64
npad = nt * 64
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
A = 2 << 20
up = lambda x: (x + A - 1) // A * A
sizes = [npad * L] * 4 + [npad * LB] + [npad * L] * 2 + [npad * 2] * 2 + [npad * 4]           # seq0 qual0 seq1 qual1 bc out0 out1 lk0 lk1 assign
arena = torch.empty(sum(up(s) for s in sizes) + A, dtype=torch.uint8, device=dev)
blk_arena = torch.empty(up(nt * lay.in_block) + up(nt * lay.out_block) + A, dtype=torch.uint8, device=dev)
ctx = seqkit_amd.Context(0)
ctx.set_barcodes(table, 1)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
off = (arena.data_ptr() + A - 1) // A * A - arena.data_ptr()
cut = []
for s in sizes:
    cut.append(arena[off:off + s])
    off += up(s)
seq = [cut[0].view(npad, L), cut[2].view(npad, L)]
qual = [cut[1].view(npad, L), cut[3].view(npad, L)]
bc = cut[4].view(npad, LB)
out = [cut[5].view(npad, L), cut[6].view(npad, L)]
lk = [cut[7].view(torch.int16), cut[8].view(torch.int16)]
assign = cut[9].view(torch.int32)
bench.gen_shard(torch, dev, npad, table, seed=4000, chunk=2_000_000, into=(seq, qual, bc))
boff = (blk_arena.data_ptr() + A - 1) // A * A - blk_arena.data_ptr()
bin_ = blk_arena[boff:boff + nt * lay.in_block]
bout = blk_arena[boff + up(nt * lay.in_block):boff + up(nt * lay.in_block) + nt * lay.out_block]
bench.pack_blocked(torch, lay, seq, qual, bc, nt, dst=bin_.view(nt, lay.in_block))
# separately allocated copies
s_seq = [x.clone() for x in seq]; s_qual = [x.clone() for x in qual]; s_bc = bc.clone()
s_out = [torch.empty_like(x) for x in out]; s_lk = [torch.empty_like(x) for x in lk]; s_assign = torch.empty_like(assign)
s_bin = bin_.clone(); s_bout = torch.empty_like(bout)
torch.cuda.synchronize()


def soa(seq, qual, bc, out, lk, assign):
    mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr(), "lowest_k": lk[i].data_ptr()} for i in range(2)]
    return lambda: ctx.fused_pass_dev(n, L, 20, mates, bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr(), counts=counts.data_ptr())


cases = [("SoA, one allocation", soa(seq, qual, bc, out, lk, assign)), ("SoA, ten allocations", soa(s_seq, s_qual, s_bc, s_out, s_lk, s_assign)),
         ("blocked, one allocation", lambda: ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())),
         ("blocked, two allocations", lambda: ctx.fused_pass_blocked_dev(lay, s_bin.data_ptr(), s_bout.data_ptr(), n, 20, counts=counts.data_ptr()))]
for rnd in range(2):
    for name, run in cases:
        run(); run(); ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            run()
        ms = ctx.timer_stop() / 3
        print(f"round {rnd} {name:26s}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}%", flush=True)
