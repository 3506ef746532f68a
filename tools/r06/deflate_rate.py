#!/usr/bin/env python3
"""Rate of the device DEFLATE on FASTQ text: the kernel alone (device-resident, HIP events) and the host entry point (PCIe both ways,
framing), next to zlib level 1 / 6 on one core of this host.  usage: deflate_rate.py [MB]"""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = seqkit_amd.Context(0)
seq, qual = synth.make_reads(40000, 150, seed=31)
mu = 36.0 - 16.0 * (np.arange(150) / 149) ** 2
rng = np.random.default_rng(31)
qual = (np.clip(np.rint(rng.normal(mu, 6.0, size=seq.shape)), 2, 40) + 33).astype(np.uint8)      # cfg 1's quality model
unit = synth.fastq_text(seq, qual, prefix="SIM:31")
data = (unit * (1 + (mb << 20) // len(unit)))[:mb << 20]
B = 0xff00
n = -(-len(data) // B)
blocks = np.zeros(n, dtype=ctx.DEFLATE_BLOCK_DTYPE)
for i in range(n):
    blocks[i] = (i * B, min(B, len(data) - i * B), 0)
src = np.frombuffer(data + bytes(8), dtype=np.uint8)
d_in, d_blk = ctx.malloc_device(src.nbytes + 64), ctx.malloc_device(blocks.nbytes + 64)
d_slots, d_tok = ctx.malloc_device(n * 81920 + 64), ctx.malloc_device(n * B * 4 + 64)
d_res, d_crc = ctx.malloc_device(n * 8 + 64), ctx.malloc_device(n * 4 + 64)
ctx.copy_h2d(d_in, src); ctx.copy_h2d(d_blk, blocks.view(np.uint8)); ctx.sync()
ts = []
for _ in range(4):
    ctx.timer_start()
    ctx._check(ctx._lib.sk_bgzf_deflate_dev(ctx._h, d_in, d_blk, n, d_slots, 81920, d_tok, d_res, d_crc), "sk_bgzf_deflate_dev")
    ts.append(ctx.timer_stop())
res = np.empty(2 * n, dtype=np.uint32)
ctx.copy_d2h(res, d_res); ctx.sync()
ms = sorted(ts[1:])[1]
print(f"kernel (deflate + crc): {n} blocks, {len(data) / 1e6:.0f} MB -> {res[0::2].sum() / 1e6:.0f} MB (1 : {len(data) / res[0::2].sum():.2f}); {ms:.2f} ms = {len(data) / ms / 1e6:.1f} GB/s in")
t0 = time.perf_counter(); comp = ctx.bgzf_deflate(data); dt = time.perf_counter() - t0
t0 = time.perf_counter(); comp = ctx.bgzf_deflate(data); dt = min(dt, time.perf_counter() - t0)
print(f"host entry point (H2D, kernel, D2H of the slots, framing): {dt * 1e3:.1f} ms = {len(data) / dt / 1e9:.2f} GB/s in; members {len(comp) / 1e6:.0f} MB")
for lv in (1, 6):
    t0 = time.perf_counter(); c = zlib.compress(data[:32 << 20], lv); dt = time.perf_counter() - t0
    print(f"zlib level {lv}, one core: {(32 << 20) / dt / 1e6:.0f} MB/s, 1 : {(32 << 20) / len(c):.2f}")
