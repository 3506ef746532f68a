#!/usr/bin/env python3
"""Device-resident rate of the barcode census (row f3): rows/s and HBM fraction for dry-run-like and statistics-like inputs.
usage: python tools/census_rates.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)


def run(name, bc, L, assign=None, bytes_per_row=None):
    stride = bc.shape[1]
    rows = bc.shape[0]
    ts = []
    for _ in range(4):
        ctx.census_reset()
        ctx.sync()
        ctx.timer_start()
        ctx.census_add_dev(bc.data_ptr(), stride, L, rows, assign.data_ptr() if assign is not None else 0, 0)
        ts.append(ctx.timer_stop())
    st = ctx.census_stats()
    ms = sorted(ts[1:])[1]
    b = bytes_per_row or (stride + (4 if assign is not None else 0))
    print(f"{name:66s} {ms:8.3f} ms {rows / ms / 1e6:8.2f} G rows/s {rows * b / ms / 1e6:8.1f} GB/s ({rows * b / ms / 1e6 / 80:.1f}% of 8 TB/s)"
          f"  distinct {st['distinct']} counted {st['counted']} slots {st['slots']}", flush=True)


table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
reps = max(1, n // 1_000_000)
bc = torch.from_numpy(bc_np).to(dev).repeat(reps, 1).contiguous()
rows = bc.shape[0]
run("statistics-like: every row, per index 85 % exact, 10 % sub, 5 % random", bc, 17)
assign = torch.empty((rows,), dtype=torch.int32, device=dev)
ctx.demux_assign_dev(bc.data_ptr(), 17, rows, assign.data_ptr())
ctx.sync()
run("dry-run-like: only unassigned rows (17 B + 4 B assign)", bc, 17, assign)
g = torch.Generator(device=dev)
g.manual_seed(3)
rnd = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (rows, 16), device=dev, generator=g)].contiguous()
run("worst case: every row a new 16-mer (16 B/row)", rnd, 16)
# where does the time go: only exact sheet barcodes (everything is counted in the workgroups' LDS tables) and only
# single-substitution neighbours (6.5 k distinct keys: more than an LDS table holds)
for name, kw in (("all rows exact sheet barcodes (LDS tables only)", dict(p_exact=1.0, p_sub=0.0)),
                 ("clean run: per index 97 % exact, 2.5 % one substitution, 0.5 % random", dict(p_exact=0.97, p_sub=0.025)),
                 ("exact + single substitutions, no random halves", dict(p_exact=0.85, p_sub=0.15))):
    b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
    b = torch.from_numpy(b_np).to(dev).repeat(reps, 1).contiguous()
    run(name, b, 17)
