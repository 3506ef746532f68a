#!/usr/bin/env python3
"""The row-major matrices of the fused pass carved from ONE allocation in different ways (order, alignment, padding between
the arrays), each timed on the same data in one process.  usage: python tools/arena_sweep.py [clusters]"""
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
table = synth.make_sheet(96, 8, dual=True, seed=4)
npad = (n + 63) // 64 * 64
names = ["seq0", "qual0", "seq1", "qual1", "bc", "out0", "out1", "lk0", "lk1", "assign"]
size = {"seq0": npad * L, "qual0": npad * L, "seq1": npad * L, "qual1": npad * L, "bc": npad * LB, "out0": npad * L, "out1": npad * L,
        "lk0": npad * 2, "lk1": npad * 2, "assign": npad * 4}
arena = torch.empty(sum(size.values()) + (4 << 30), dtype=torch.uint8, device=dev)
ctx = seqkit_amd.Context(0)
ctx.set_barcodes(table, 1)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
ref_seq, ref_qual, ref_bc = bench.gen_shard(torch, dev, npad, table, seed=4000, chunk=2_000_000)
ref = {"seq0": ref_seq[0].view(-1), "qual0": ref_qual[0].view(-1), "seq1": ref_seq[1].view(-1), "qual1": ref_qual[1].view(-1), "bc": ref_bc.view(-1)}
base0 = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()


def carve(order, align, pad):
    off = base0
    at = {}
    for k, nm in enumerate(order):
        off = (off + align - 1) // align * align
        at[nm] = off
        off += size[nm] + (pad * (k + 1) if pad else 0)
    return at


schemes = [("in-order, 2 MiB aligned", names, 2 << 20, 0), ("in-order, 4 KiB aligned", names, 4096, 0),
           ("per mate: q s out", ["qual0", "seq0", "out0", "qual1", "seq1", "out1", "bc", "lk0", "lk1", "assign"], 2 << 20, 0),
           ("in-order, 2 MiB + k*64 KiB stagger", names, 2 << 20, 65536),
           ("in-order, 1 GiB aligned", names, 1 << 30, 0), ("in-order, 2 MiB aligned (again)", names, 2 << 20, 0)]
# controls on the same unit: the matrices as separate allocations, and the tile-blocked pair from one allocation
c_out = [torch.empty_like(ref_seq[0]) for _ in range(2)]
c_lk = [torch.empty((npad,), dtype=torch.int16, device=dev) for _ in range(2)]
c_assign = torch.empty((npad,), dtype=torch.int32, device=dev)
c_mates = [{"seq": ref_seq[i].data_ptr(), "qual": ref_qual[i].data_ptr(), "len": 0, "out_seq": c_out[i].data_ptr(), "lowest_k": c_lk[i].data_ptr()} for i in range(2)]
from seqkit_amd import capi  # noqa: E402
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
nt = npad // 64
blk = torch.empty(nt * (lay.in_block + lay.out_block) + (4 << 20), dtype=torch.uint8, device=dev)
b0 = (blk.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - blk.data_ptr()
b_in = blk[b0:b0 + nt * lay.in_block]
b_out = blk[b0 + nt * lay.in_block:b0 + nt * (lay.in_block + lay.out_block)]
bench.pack_blocked(torch, lay, ref_seq, ref_qual, ref_bc, nt, dst=b_in.view(nt, lay.in_block))
torch.cuda.synchronize()
for name, run in (("control: matrices as ten allocations", lambda: ctx.fused_pass_dev(n, L, 20, c_mates, bc=ref_bc.data_ptr(), bc_stride=LB, assign=c_assign.data_ptr(), counts=counts.data_ptr())),
                  ("control: blocked pair, one allocation", lambda: ctx.fused_pass_blocked_dev(lay, b_in.data_ptr(), b_out.data_ptr(), n, 20, counts=counts.data_ptr()))):
    ts = []
    for _ in range(2):
        run(); run(); ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            run()
        ts.append(ctx.timer_stop() / 3)
    print(f"{name:42s}: {ts[0]:7.3f} {ts[1]:7.3f} ms  {925 * n / min(ts) / 1e6 / 80:.1f}%", flush=True)
for name, order, align, pad in schemes:
    at = carve(order, align, pad)
    for nm in ("seq0", "qual0", "seq1", "qual1", "bc"):
        arena[at[nm]:at[nm] + size[nm]].copy_(ref[nm])
    torch.cuda.synchronize()
    p = {nm: arena.data_ptr() + at[nm] for nm in names}
    mates = [{"seq": p[f"seq{i}"], "qual": p[f"qual{i}"], "len": 0, "out_seq": p[f"out{i}"], "lowest_k": p[f"lk{i}"]} for i in range(2)]
    run = lambda: ctx.fused_pass_dev(n, L, 20, mates, bc=p["bc"], bc_stride=LB, assign=p["assign"], counts=counts.data_ptr())
    ts = []
    for _ in range(2):
        run(); run(); ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            run()
        ts.append(ctx.timer_stop() / 3)
    print(f"{name:42s}: {ts[0]:7.3f} {ts[1]:7.3f} ms  {925 * n / min(ts) / 1e6 / 80:.1f}%", flush=True)
