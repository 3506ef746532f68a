#!/usr/bin/env python3
"""A small fixed workload for rocprofv3 --pmc passes: the fused pass on one device-resident shard.
usage: python3 tools/prof_fused.py [pairs] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
seq, qual, bc = bench.gen_shard(torch, dev, n, table, seed=1, chunk=2_000_000)
out = [torch.empty_like(seq[0]) for _ in range(2)]
lk = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]
assign = torch.empty((n,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr(),
          "lowest_k": lk[i].data_ptr()} for i in range(2)]
for _ in range(iters):
    ctx.fused_pass_dev(n, 150, 20, mates, bc=bc.data_ptr(), bc_stride=17, assign=assign.data_ptr())
ctx.sync()
print("pairs", n, "iters", iters)
