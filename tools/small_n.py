#!/usr/bin/env python3
"""Tile-pass shape (workgroups of four waves per CU) against problem size: small inputs want more resident waves.
usage: python tools/small_n.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402

dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(7)
for n in (250_000, 1_000_000, 4_000_000, 16_000_000):
    q = torch.randint(35, 74, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    s = torch.randint(65, 85, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    o = torch.empty_like(s)
    lk = torch.empty((n,), dtype=torch.int16, device=dev)
    mate_t = [{"seq": 0, "qual": q.data_ptr(), "len": 0, "out_seq": 0, "lowest_k": lk.data_ptr()}]
    mate_mt = [{"seq": s.data_ptr(), "qual": q.data_ptr(), "len": 0, "out_seq": o.data_ptr(), "lowest_k": lk.data_ptr()}]
    for name, mates in (("trim", mate_t), ("mask+trim", mate_mt)):
        row = []
        for wg in ("2", "3", "4", "6", "8"):
            os.environ["SK_TILE_WAVES"], os.environ["SK_TILE_WGS"] = "4", wg
            for _ in range(3):
                ctx.fused_pass_dev(n, 150, 20, mates)
            ctx.sync()
            ctx.timer_start()
            for _ in range(20):
                ctx.fused_pass_dev(n, 150, 20, mates)
            row.append(ctx.timer_stop() / 20)
        print(f"n={n:9d} {name:10s} WGs/CU 2,3,4,6,8: " + "  ".join(f"{ms * 1e3:8.1f} us" for ms in row), flush=True)
