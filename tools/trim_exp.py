#!/usr/bin/env python3
"""Trim by quality alone (cfg 2) on device-resident qualities of several shapes, this build against other builds of the
same C-ABI in one process.  usage: SK_LIBS=tools/ab/r01.so python tools/trim_exp.py [reads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
g = torch.Generator(device=dev)
g.manual_seed(7)
mu = 36.0 - 16.0 * (torch.arange(150, device=dev, dtype=torch.float32) / 149) ** 2


def gen(kind):
    if kind == "uniform Q2-Q40":
        return torch.randint(35, 74, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    if kind == "every row all '#' (no scan ever breaks)":
        return torch.full((n, 150), ord("#"), dtype=torch.uint8, device=dev)
    if kind == "uniform Q2-Q30 (drifts down: long scans)":
        return torch.randint(35, 64, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    q = torch.empty((n, 150), dtype=torch.uint8, device=dev)
    for r0 in range(0, n, 2_000_000):
        m = min(2_000_000, n - r0)
        x = torch.randn((m, 150), generator=g, device=dev) * 6.0 + mu
        q[r0:r0 + m] = (x.round_().clamp_(2, 40) + 33).to(torch.uint8)
    if kind == "read-like + 5% all-'#' rows":
        bad = torch.rand((n,), generator=g, device=dev) < 0.05
        q[bad] = ord("#")
    return q


lk = torch.empty((n,), dtype=torch.int16, device=dev)
for kind in ("uniform Q2-Q40", "every row all '#' (no scan ever breaks)", "uniform Q2-Q30 (drifts down: long scans)", "read-like", "read-like + 5% all-'#' rows"):
    q = gen(kind)
    exp = orc.trim_batch(q[:200_000].cpu().numpy(), None, 20)
    for name, ctx in ctxs:
        def run():
            ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, 20, lk.data_ptr())
        run(); run(); ctx.sync()
        assert np.array_equal(lk[:200_000].cpu().numpy().view(np.uint16), exp), (kind, name)
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10):
                run()
            ts.append(ctx.timer_stop() / 10)
        ms = sorted(ts)[2]
        print(f"{kind:42s} {name:5s}: {ms:7.4f} ms  {n / ms / 1e6:6.2f} G reads/s  {152 * n / ms / 1e6 / 80:5.1f}% of 8 TB/s", flush=True)
    del q
