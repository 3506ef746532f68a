#!/usr/bin/env python3
"""One shape of the fused pass, a few launches: the thing to put under rocprofv3 (tools/profile_cmd.sh).
usage: python tools/fused_one.py [single|single_ragged|paired_detail] [reads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "single"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000_000
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(7)
mu = 36.0 - 16.0 * (torch.arange(150, device=dev, dtype=torch.float32) / 149) ** 2


def reads():
    q = torch.empty((n, 150), dtype=torch.uint8, device=dev)
    for r0 in range(0, n, 2_000_000):
        m = min(2_000_000, n - r0)
        q[r0:r0 + m] = ((torch.randn((m, 150), generator=g, device=dev) * 6.0 + mu).round_().clamp_(2, 40) + 33).to(torch.uint8)
    s = torch.randint(65, 85, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    return s, q, torch.empty_like(s), torch.empty((n,), dtype=torch.int16, device=dev)


paired = kind == "paired_detail"
table = synth.make_sheet(96, 8, dual=True, seed=4) if paired else synth.make_sheet(16, 8, dual=False, seed=3)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4 if paired else 3, halves=2 if paired else 1)
bc = torch.from_numpy(bc_np).to(dev).repeat(n // 1_000_000, 1).contiguous()
mates, keep = [], []
for _ in range(2 if paired else 1):
    s, q, o, lk = reads()
    keep.append((s, q, o, lk))
    mates.append({"seq": s.data_ptr(), "qual": q.data_ptr(), "len": 0, "out_seq": o.data_ptr(), "lowest_k": lk.data_ptr()})
if kind == "single_ragged":
    ln = torch.randint(100, 151, (n,), dtype=torch.int16, device=dev, generator=g)
    mates[0]["len"] = ln.data_ptr()
assign = torch.empty((n,), dtype=torch.int32, device=dev)
cnt = torch.zeros((table.shape[0] + 3,), dtype=torch.int64, device=dev)
kw = dict(bc=bc.data_ptr(), bc_stride=bc.shape[1], assign=assign.data_ptr(), counts=cnt.data_ptr())
if paired:
    low = torch.empty((n,), dtype=torch.uint8, device=dev)
    first = torch.empty((n,), dtype=torch.int16, device=dev)
    last = torch.empty((n,), dtype=torch.int16, device=dev)
    kw.update(lowest_diff=low.data_ptr(), first_idx=first.data_ptr(), last_idx=last.data_ptr())
    ctx.set_detail_mode(capi.SK_DETAIL_MATCHED)
bpu = {"single": 464, "single_ragged": 466, "paired_detail": 930}[kind]
torch.cuda.synchronize()
for _ in range(3):
    ctx.fused_pass_dev(n, 150, 20, mates, **kw)
ctx.sync()
ctx.timer_start()
for _ in range(5):
    ctx.fused_pass_dev(n, 150, 20, mates, **kw)
ms = ctx.timer_stop() / 5
print(f"fused {kind} n={n}: {ms:.4f} ms  {bpu * n / ms / 1e6:.1f} GB/s  {bpu * n / ms / 1e6 / 80:.1f}% of 8 TB/s ({bpu} B per read / cluster)")
