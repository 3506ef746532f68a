#!/usr/bin/env python3
"""One shape of trim by quality alone, a few launches: the thing to put under rocprofv3 (tools/profile_cmd.sh).
usage: [SK_LIB=tools/ab/x.so] python tools/trim_one.py [uniform|readlike|readlike5] [reads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000_000
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(os.environ["SK_LIB"]) if os.environ.get("SK_LIB") else None)
g = torch.Generator(device=dev)
g.manual_seed(7)
if kind == "uniform":
    q = torch.randint(35, 74, (n, 150), dtype=torch.uint8, device=dev, generator=g)
else:
    mu = 36.0 - 16.0 * (torch.arange(150, device=dev, dtype=torch.float32) / 149) ** 2
    q = torch.empty((n, 150), dtype=torch.uint8, device=dev)
    for r0 in range(0, n, 2_000_000):
        m = min(2_000_000, n - r0)
        q[r0:r0 + m] = ((torch.randn((m, 150), generator=g, device=dev) * 6.0 + mu).round_().clamp_(2, 40) + 33).to(torch.uint8)
if kind == "readlike5":
    bad = torch.rand((n,), generator=g, device=dev) < 0.05
    q[bad] = ord("#")
lk = torch.empty((n,), dtype=torch.int16, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, 20, lk.data_ptr())
ctx.sync()
ctx.timer_start()
for _ in range(5):
    ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, 20, lk.data_ptr())
ms = ctx.timer_stop() / 5
print(f"{kind} n={n}: {ms:.4f} ms  {152 * n / ms / 1e6 / 80:.1f}% of 8 TB/s")
