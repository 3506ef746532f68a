#!/usr/bin/env python3
"""`sam count` of several builds of the C-ABI on the same 100 M coordinate-sorted records in one process.
usage: SK_LIBS=tools/ab/x.so python tools/count_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqkit_amd  # noqa: E402

dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
g = torch.Generator(device=dev)
g.manual_seed(7)
n, n_chr = 100_000_000, 24
rng = np.random.default_rng(9)
for order in ("coordinate-sorted", "positions in random order"):
    tid64 = torch.randint(0, n_chr, (n,), dtype=torch.int64, device=dev, generator=g)
    pos64 = torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g)
    if order == "coordinate-sorted":
        key = torch.sort((tid64 << 32) | pos64).values
        tid64, pos64 = key >> 32, key & 0xffffffff
        del key
    else:
        tid64 = torch.sort(tid64).values
    ctid, cpos = tid64.to(torch.int32).contiguous(), pos64.to(torch.int32).contiguous()
    del tid64, pos64
    ctl = torch.randint(50, 600, (n,), dtype=torch.int32, device=dev, generator=g)
    cflag = torch.full((n,), 99, dtype=torch.int16, device=dev)
    cmapq = torch.full((n,), 60, dtype=torch.uint8, device=dev)
    cmpos = (cpos + ctl // 2).contiguous()
    rchr = np.sort(rng.integers(0, n_chr, size=20000)).astype(np.int32)
    rstart = rng.integers(0, 100_000_000, size=20000).astype(np.uint32)
    chr_off = np.searchsorted(rchr, np.arange(n_chr + 1)).astype(np.int32)
    ref = None
    for name, ctx in ctxs:
        ctx.count_set_regions(chr_off, rstart, (rstart + 1000).astype(np.uint32))
        def run():
            ctx.count_add_dev(cflag.data_ptr(), cmapq.data_ptr(), ctid.data_ptr(), ctid.data_ptr(), cpos.data_ptr(), cmpos.data_ptr(), ctl.data_ptr(), 0, n)
        run(); ctx.sync()
        got = ctx.count_get().copy()
        if ref is None:
            ref = got
        assert np.array_equal(got, ref), name
        ts = []
        for _ in range(3):
            ctx.timer_start()
            for _ in range(3):
                run()
            ts.append(ctx.timer_stop() / 3)
        ms = sorted(ts)[1]
        print(f"{order:28s} {name:6s}: {ms:7.3f} ms  {n / ms / 1e6:6.1f} G records/s  {31 * n / ms / 1e6 / 80:5.1f}% of 8 TB/s", flush=True)
    del ctid, cpos, ctl, cflag, cmapq, cmpos
