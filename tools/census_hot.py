#!/usr/bin/env python3
"""Front-table only shapes of the census: every row one of K equally likely barcodes (K <= what a front table holds).
usage: census_hot.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(3)
alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
for K in (1, 8, 96, 400, 1200):
    keys = alpha[torch.randint(0, 4, (K, 17), device=dev, generator=g)]
    bc = keys[torch.randint(0, K, (n,), device=dev, generator=g)].contiguous()
    ts = []
    for _ in range(4):
        ctx.census_reset()
        ctx.sync()
        ctx.timer_start()
        ctx.census_add_dev(bc.data_ptr(), 17, 17, n, 0, 0)
        ts.append(ctx.timer_stop())
    st = ctx.census_stats()
    print(f"K={K:5d}: {sorted(ts[1:])[1]:.3f} ms  {n / sorted(ts[1:])[1] / 1e6:.1f} G rows/s  distinct {st['distinct']}", flush=True)
