import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(16, 8, dual=False, seed=3)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3)
for reps in (10, 100, 400):
    n = 1_000_000 * reps
    bc = torch.from_numpy(bc_np).to(dev).repeat(reps, 1).contiguous()
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    for _ in range(2): ctx.demux_assign_dev(bc.data_ptr(), 8, n, assign.data_ptr())
    ctx.sync(); ctx.timer_start()
    for _ in range(5): ctx.demux_assign_dev(bc.data_ptr(), 8, n, assign.data_ptr())
    ms = ctx.timer_stop() / 5
    print(f"n={n}: {ms:.4f} ms  {n/ms/1e6:.1f} G reads/s  {12*n/ms/1e6:.0f} GB/s")
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
for reps in (10, 100):
    n = 1_000_000 * reps
    bc = torch.from_numpy(bc_np).to(dev).repeat(reps, 1).contiguous()
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    for _ in range(2): ctx.demux_assign_dev(bc.data_ptr(), 17, n, assign.data_ptr())
    ctx.sync(); ctx.timer_start()
    for _ in range(5): ctx.demux_assign_dev(bc.data_ptr(), 17, n, assign.data_ptr())
    ms = ctx.timer_stop() / 5
    print(f"96 dual-index n={n}: {ms:.4f} ms  {n/ms/1e6:.1f} G pairs/s  {21*n/ms/1e6:.0f} GB/s")
