#!/usr/bin/env python3
"""How much does WHERE a buffer was allocated move the fused pass?  K sets of buffers for the same shard, allocated one after
the other in one process and all kept, the kernel timed on each — for the tile-blocked layout (one input + one output
buffer) and for the row-major matrices (nine).  usage: python tools/placement_draws.py [clusters] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
L, LB = 150, 17
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
nt = (n + 63) // 64
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
blk = [(torch.empty(nt * lay.in_block, dtype=torch.uint8, device=dev), torch.empty(nt * lay.out_block, dtype=torch.uint8, device=dev)) for _ in range(K)]
ctx = seqkit_amd.Context(0)
ctx.set_barcodes(table, 1)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
vin = blk[0][0].view(nt, lay.in_block)
gen_chunk = 2_000_000 // 64 * 64
bench.gen_shard(torch, dev, nt * 64, table, seed=4000, chunk=gen_chunk,
                sink=lambda r0, cs, cq, b: bench.pack_blocked(torch, lay, cs, cq, b, b.shape[0] // 64, dst=vin[r0 // 64:(r0 + b.shape[0]) // 64]))
for k in range(1, K):
    blk[k][0].copy_(blk[0][0])
torch.cuda.synchronize()


def probe(run, reps=3):
    run(); run(); ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(reps):
            run()
        ts.append(ctx.timer_stop() / reps)
    return sorted(ts)[1]


for rnd in range(2):
    for k in range(K):
        ms = probe(lambda: ctx.fused_pass_blocked_dev(lay, blk[k][0].data_ptr(), blk[k][1].data_ptr(), n, 20, counts=counts.data_ptr()))
        print(f"round {rnd} blocked set {k}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}%   in @ {blk[k][0].data_ptr():#x}", flush=True)
# the row-major matrices: unpack set 0 into K sets of nine arrays
del blk[1:]
torch.cuda.empty_cache()
u8 = torch.uint8
sets = []
v = blk[0][0].view(nt, lay.in_block)
row = 64 * L
for k in range(min(K, 3)):
    seq = [v[:, lay.in_seq[i]:lay.in_seq[i] + row].reshape(nt * 64, L).contiguous() for i in range(2)]
    qual = [v[:, lay.in_qual[i]:lay.in_qual[i] + row].reshape(nt * 64, L).contiguous() for i in range(2)]
    bc = v[:, lay.in_bc:lay.in_bc + 64 * LB].reshape(nt * 64, LB).contiguous()
    out = [torch.empty_like(seq[0]) for _ in range(2)]
    lk = [torch.empty((nt * 64,), dtype=torch.int16, device=dev) for _ in range(2)]
    assign = torch.empty((nt * 64,), dtype=torch.int32, device=dev)
    sets.append((seq, qual, bc, out, lk, assign))
torch.cuda.synchronize()
for rnd in range(2):
    for k, (seq, qual, bc, out, lk, assign) in enumerate(sets):
        mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr(), "lowest_k": lk[i].data_ptr()} for i in range(2)]
        ms = probe(lambda: ctx.fused_pass_dev(n, L, 20, mates, bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr(), counts=counts.data_ptr()))
        print(f"round {rnd} SoA set {k}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}%", flush=True)
