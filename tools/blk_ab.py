#!/usr/bin/env python3
"""The tile-blocked fused pass of several builds of the library (SK_LIBS=a.so,b.so: compile-time variants) on the same
buffers of one GPU unit, timed alternately.  usage: SK_LIBS=tools/ab/x.so,... python tools/blk_ab.py [clusters] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L, LB = 150, 17
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = []
for name, path in libs:
    c = seqkit_amd.Context(0, lib_path=path)
    c.set_barcodes(table, 1)
    ctxs.append((name, c))
nt = (n + 63) // 64
seq, qual, bc = bench.gen_shard(torch, dev, nt * 64, table, seed=4000, chunk=2_000_000)
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
bin_, bout = bench.pack_blocked(torch, lay, seq, qual, bc, nt)
del seq, qual, bc
torch.cuda.empty_cache()
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ref = None
for r in range(rounds):
    for name, ctx in ctxs:
        def run():
            ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())
        run(); run(); ctx.sync()
        if r == 0:
            if ref is None:
                ref = bout.clone()
            else:
                assert torch.equal(ref, bout), name
        ts = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(3):
                run()
            ts.append(ctx.timer_stop() / 3)
        ms = sorted(ts)[1]
        print(f"round {r} {name:12s}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
