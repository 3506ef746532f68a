#!/usr/bin/env python3
"""Small census launches (the command lines' batches): time per launch against rows, two shapes.  usage: python tools/census_small.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(os.environ["SK_LIB"]) if os.environ.get("SK_LIB") else None)
table = synth.make_sheet(96, 8, dual=True, seed=4)
for case, kw in (("exact", dict(p_exact=1.0, p_sub=0.0)), ("noisy", {})):
    b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
    bc = torch.from_numpy(b_np).to(dev).repeat(8, 1).contiguous()
    for n in (64_000, 250_000, 1_000_000, 4_000_000, 8_000_000):
        ts = []
        for _ in range(7):
            ctx.census_reset(); ctx.sync(); ctx.timer_start()
            ctx.census_add_dev(bc.data_ptr(), 17, 17, n, 0, 0)
            ts.append(ctx.timer_stop())
        print(f"{case:6s} {n:9d} rows: {sorted(ts)[3] * 1e3:8.1f} us", flush=True)
