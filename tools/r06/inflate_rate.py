#!/usr/bin/env python3
"""Device-resident rate of bgzf_inflate_kernel (+ CRC) on BAM-like blocks.  usage: inflate_rate.py [MB of inflated data per shape] [lib]
shapes: `random` = tools/bam_e2e.py's records (random bases and qualities: literals, 1.5 : 1), `sorted` = reads drawn from a small genome in
position order with binned qualities (what a coordinate-sorted BAM of a modern instrument looks like: matches, 4 : 1)."""
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

import seqkit_amd  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = sys.argv[2] if len(sys.argv) > 2 else None
ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(lib)) if lib else seqkit_amd.Context(0)
rng = np.random.default_rng(3)
codes = np.array([1, 2, 4, 8], dtype=np.uint8)


def records(kind, n):
    out = bytearray()
    genome = codes[rng.integers(0, 4, size=200_000)]
    qbins = np.array([2, 12, 23, 37], dtype=np.uint8)
    for i in range(n):
        tl = int(rng.lognormal(np.log(170), 0.35))
        name = b"A00123:45:HXXXXXXX:1:%d:%d:%d\0" % (1101 + i // 5000, 1000 + (i * 7) % 30000, 1000 + (i * 13) % 30000)
        if kind == "random":
            nib = codes[rng.integers(0, 4, size=150)]
            q = rng.integers(2, 41, size=150, dtype=np.uint8)
        else:
            p = (i * 3) % (len(genome) - 150)
            nib = genome[p:p + 150].copy()
            if rng.random() < 0.3:
                nib[int(rng.integers(0, 150))] = codes[int(rng.integers(0, 4))]
            q = qbins[np.minimum(3, rng.geometric(0.75, size=150) - 1)][::-1].copy()
            q[:100] = 37
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        flag = 99 if i % 2 == 0 else 147
        body = struct.pack("<iiBBHHHiiii", 0, i * 3, len(name), 60, 4680, 1, flag, 150, 0, i * 3 + tl, tl if i % 2 == 0 else -tl) + name + struct.pack("<I", 150 << 4) + packed + q.tobytes()
        out += struct.pack("<i", len(body)) + body
    return bytes(out)


for kind in ("random", "sorted"):
    unit = records(kind, 20000)
    raws = [unit[o:o + 65280] for o in range(0, len(unit) - 65280, 65280)]
    comps = [zlib.compressobj(6 if kind == "sorted" else 1, zlib.DEFLATED, -15) for _ in raws]
    pays = [c.compress(r) + c.flush() for c, r in zip(comps, raws)]
    reps = max(1, (mb << 20) // (len(raws) * 65280))
    n = len(raws) * reps
    blocks = np.zeros(n, dtype=ctx.BGZF_BLOCK_DTYPE)
    comp = bytearray()
    offs = []
    for p in pays:
        comp += b"\0" * 18
        offs.append(len(comp))
        comp += p + b"\0" * 8
    comp_one = bytes(comp)
    out_off = 0
    for r in range(reps):
        for j, (p, raw) in enumerate(zip(pays, raws)):
            blocks[r * len(raws) + j] = (r * len(comp_one) + offs[j], len(p), len(raw), out_off, zlib.crc32(raw) & 0xFFFFFFFF, 0)
            out_off += len(raw)
    comp_all = np.frombuffer(comp_one * reps + bytes(64), dtype=np.uint8)
    d_comp, d_blocks = ctx.malloc_device(comp_all.nbytes + 64), ctx.malloc_device(blocks.nbytes + 64)
    d_out, d_status = ctx.malloc_device(out_off + 64), ctx.malloc_device(4 * n + 64)
    ctx.copy_h2d(d_comp, comp_all); ctx.copy_h2d(d_blocks, blocks.view(np.uint8)); ctx.sync()
    for crc in (False, True):
        ts = []
        for _ in range(4):
            ctx.timer_start()
            ctx.bgzf_inflate_dev(d_comp, d_blocks, n, d_out, d_status, crc)
            ts.append(ctx.timer_stop())
        st = np.empty(n, dtype=np.uint32)
        ctx.copy_d2h(st, d_status); ctx.sync()
        ms = sorted(ts[1:])[1]
        print(f"{kind:7s} crc={int(crc)}: {n} blocks, {len(comp_all) / 1e6:.0f} MB -> {out_off / 1e6:.0f} MB (1 : {out_off / len(comp_all):.2f}); {ms:.2f} ms = "
              f"{out_off / ms / 1e6:.1f} GB/s inflated, {len(comp_all) / ms / 1e6:.1f} GB/s compressed; {n / ms / 1e3:.2f} M blocks/s; bad status {int((st != 0).sum())}", flush=True)
    if hasattr(ctx._lib, "sk_debug_inflate_stamps"):                 # a -DSK_INF_STAMPS build: shader cycles per phase of the symbol loop, summed over the waves
        import ctypes as C
        acc = (C.c_ulonglong * 16)()
        ctx._lib.sk_debug_inflate_stamps(acc, 1)
        ctx.bgzf_inflate_dev(d_comp, d_blocks, n, d_out, d_status, False); ctx.sync()
        ctx._lib.sk_debug_inflate_stamps(acc, 1)
        names = {0: "-", 1: "the group loop (refill, per-lane decode, chain, places, stores)", 2: "-", 3: "a symbol bit by bit (long codes, end of block, a match across the unit's end)",
                 4: "batch: matches from flushed bytes", 5: "batch: matches through the ring", 6: "flush", 7: "a match across the unit's end: its copy", 8: "(into finish_batch)", 9: "(into flush)"}
        blocks_done, total = acc[14], acc[15]
        print(f"   stamps: {blocks_done} blocks, {total / max(1, blocks_done):.0f} cycles per block")
        inside = sum(acc[i] for i in range(10))
        print(f"   outside the symbol loop (headers, tables, final flush): {(total - inside) / max(1, blocks_done):.0f} cycles per block = {100 * (total - inside) / max(1, total):.1f} %")
        for i in (1, 3, 4, 5, 6, 7, 8, 9):
            print(f"   {names[i]:80s} {acc[i] / max(1, blocks_done):10.0f} cycles per block  {100 * acc[i] / max(1, total):5.1f} %")
    for p in (d_comp, d_blocks, d_out, d_status):
        ctx.free_device(p)
