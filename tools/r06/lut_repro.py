#!/usr/bin/env python3
"""is the 96 dual-index lookup's 400-509 us on identical bytes placement, or something else?
(1) ONE pair of buffers, 60 launches one behind the other, each timed with its own event pair: the series;
(2) K candidates for bc and K for assign, every pair timed (median of 7): the matrix;
(3) the series again on the best and on the worst pair.
usage: lut_repro.py [rows] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
ctx.set_barcodes(table, 1)
bcs = [torch.from_numpy(bc_np).to(dev).repeat(n // 1_000_000, 1).contiguous() for _ in range(K)]
asg = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(K)]
torch.cuda.synchronize()
stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)


def series(i, j, reps):
    evs = []
    with torch.cuda.stream(stream):
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ctx.demux_assign_dev(bcs[i].data_ptr(), 17, n, asg[j].data_ptr())
            e1.record(stream)
            evs.append((e0, e1))
    ctx.sync()
    return np.array([a.elapsed_time(b) * 1e3 for a, b in evs])


series(0, 0, 5)
s = series(0, 0, 60)
print(f"(1) one pair, 60 launches back to back [us]: min {s.min():.1f} median {np.median(s):.1f} max {s.max():.1f} std {s.std():.1f}")
print("    " + " ".join(f"{x:.0f}" for x in s))
m = np.zeros((K, K))
for i in range(K):
    for j in range(K):
        m[i, j] = np.median(series(i, j, 7))
print(f"(2) median of 7 per (bc candidate, assign candidate) [us]; addresses bc {[hex(b.data_ptr()) for b in bcs]} assign {[hex(a.data_ptr()) for a in asg]}")
for i in range(K):
    print("    " + " ".join(f"{m[i, j]:7.1f}" for j in range(K)))
bi, bj = np.unravel_index(m.argmin(), m.shape)
wi, wj = np.unravel_index(m.argmax(), m.shape)
for name, (i, j) in (("best", (bi, bj)), ("worst", (wi, wj))):
    s = series(i, j, 30)
    print(f"(3) {name} pair ({i},{j}), 30 launches [us]: min {s.min():.1f} median {np.median(s):.1f} max {s.max():.1f} std {s.std():.1f}")
gb = n * 21 / 1e3
print(f"    best median {m.min():.1f} us = {gb / m.min() / 8000:.3f} of 8 TB/s, worst {m.max():.1f} us = {gb / m.max() / 8000:.3f}; spread {100 * (m.max() / m.min() - 1):.1f} %")
