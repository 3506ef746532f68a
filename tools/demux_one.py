#!/usr/bin/env python3
"""Demultiplex alone on one sheet, a few launches: the thing to put under rocprofv3 (tools/profile_cmd.sh).
usage: python tools/demux_one.py [dual|dual384|cfg3] [rows] [detail]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "dual"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
detail = len(sys.argv) > 3
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
if kind == "dual":
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
elif kind == "dual384":                       # four plates: the sheet is looked up half by half (sk_lut.h)
    table = synth.make_sheet(384, 8, dual=True, seed=384)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
else:
    table = synth.make_sheet(16, 8, dual=False, seed=3)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3)
L = table.shape[1]
ctx.set_barcodes(table, 1)
if detail:
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
bc = torch.from_numpy(bc_np).to(dev).repeat(max(1, n // 1_000_000), 1)[:n].contiguous()
assign = torch.empty((n,), dtype=torch.int32, device=dev)
low = torch.empty((n,), dtype=torch.uint8, device=dev)
first = torch.empty((n,), dtype=torch.int16, device=dev)
last = torch.empty((n,), dtype=torch.int16, device=dev)
torch.cuda.synchronize()


def run():
    if detail:
        ctx.demux_assign_dev(bc.data_ptr(), L, n, assign.data_ptr(), low.data_ptr(), first.data_ptr(), last.data_ptr())
    else:
        ctx.demux_assign_dev(bc.data_ptr(), L, n, assign.data_ptr())


# Warm-up: the device's memory-side clocks ramp for the first ~20 ms of work after idle — the same launch on the same bytes takes
# 450-470 us at first and 390-395 us from the fiftieth launch on (tools/r06/lut_repro.py; profiles/r06_lut_repro.txt).  Round 5's
# kernel trace of this script (3 warm-up launches, 13 traced: 400-509 us) sat inside that ramp.  SK_ONE_WARMUP overrides.
warm = int(os.environ.get("SK_ONE_WARMUP", "150" if n >= 50_000_000 else "600"))
for _ in range(warm):
    run()
ctx.sync()
ctx.timer_start()
for _ in range(10):
    run()
ms = ctx.timer_stop() / 10
b = L + 4 + (5 if detail else 0)
print(f"{kind} n={n} detail={detail}: {ms:.4f} ms  {n / ms / 1e6:.1f} G rows/s  {n * b / ms / 1e6 / 80:.1f}% of 8 TB/s")
