#!/usr/bin/env python3
"""Input and output buffer of the blocked pass carved from ONE allocation, the output at a swept distance behind the input:
is the fast / slow pairing a function of the distance?  usage: python tools/pair_offset_exp.py [clusters]"""
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
lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
in_bytes, out_bytes = nt * lay.in_block, nt * lay.out_block
slack = 1 << 30
arena = torch.empty(in_bytes + out_bytes + slack + (2 << 20), dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()      # 2 MiB aligned start
bin_ = arena[base:base + in_bytes]
ctx = seqkit_amd.Context(0)
ctx.set_barcodes(table, 1)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
vin = bin_.view(nt, lay.in_block)
gen_chunk = 2_000_000 // 64 * 64
bench.gen_shard(torch, dev, nt * 64, table, seed=4000, chunk=gen_chunk,
                sink=lambda r0, cs, cq, b: bench.pack_blocked(torch, lay, cs, cq, b, b.shape[0] // 64, dst=vin[r0 // 64:(r0 + b.shape[0]) // 64]))
torch.cuda.synchronize()
out0 = (base + in_bytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
deltas = [0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20,
          64 << 20, 128 << 20, 256 << 20, 512 << 20, 3 << 20, 5 << 20, 12345 * 128, 1 << 30]
for rnd in range(2):
    for d in deltas:
        bout = arena[out0 + d:out0 + d + out_bytes]
        run = lambda: ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())
        run(); run(); ctx.sync()
        ctx.timer_start()
        for _ in range(3):
            run()
        ms = ctx.timer_stop() / 3
        print(f"round {rnd} out = in_end + {d:>10d}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}%", flush=True)
