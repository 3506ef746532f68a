#!/usr/bin/env python3
"""`sequence()` (sam to fastq's base decoding) of several builds of the C-ABI on the same rows in one process.
usage: SK_LIBS=tools/ab/x.so [SEQ_STRIDE=148] python tools/seq_ab.py [records]   (SEQ_STRIDE: the row pitch of qualities and output, a multiple of 4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
g = torch.Generator(device=dev)
g.manual_seed(5)
P = int(os.environ.get("SEQ_STRIDE", "152"))
L = min(150, P)
s4 = torch.randint(0, 256, (n, 76), dtype=torch.uint8, device=dev, generator=g)
q = torch.randint(0, 42, (n, P), dtype=torch.uint8, device=dev, generator=g)
ln = torch.full((n,), L, dtype=torch.int16, device=dev)
o = torch.empty((n, P), dtype=torch.uint8, device=dev)
for what in ("mixed strands", "forward only"):
    fl = (torch.randint(0, 2, (n,), dtype=torch.int16, device=dev, generator=g) * 16) if what == "mixed strands" else torch.zeros((n,), dtype=torch.int16, device=dev)
    ref = None
    for name, ctx in ctxs:
        def run():
            ctx.bam_sequence_dev(s4.data_ptr(), 76, q.data_ptr(), P, ln.data_ptr(), fl.data_ptr(), n, 10, o.data_ptr())
        run(); ctx.sync()
        got = o[:100000, :L].clone()
        if ref is None:
            ref = got
        assert torch.equal(got, ref), name
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(5):
                run()
            ts.append(ctx.timer_stop() / 5)
        ms = sorted(ts)[2]
        print(f"{what:14s} {name:8s}: {ms:7.4f} ms  {n / ms / 1e6:6.2f} G records/s  {(76 + 2 * P + 4) * n / ms / 1e6 / 80:5.1f}% of 8 TB/s (pitch {P})", flush=True)
