#!/usr/bin/env python3
"""One of the secondary kernels, warmed up and launched a few times: the thing to put under rocprofv3 (tools/profile_cmd.sh) so that every
quoted fraction has a kernel trace and PMC counters behind it (VERDICT r5 "what's weak" 6).
usage: kernel_one.py mask|bam|fragments|sequence152|sequence148|inflate_random|inflate_sorted|deflate"""
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

what = sys.argv[1]
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(7)
reps, warm = 10, 30
if what == "mask":
    n = 16_000_000
    q = torch.randint(35, 74, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    s = torch.randint(65, 85, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    o = torch.empty_like(s)
    run = lambda: ctx.mask_by_quality_dev(s.data_ptr(), q.data_ptr(), 150, n, 20, o.data_ptr())
    units, bpu = n, 450
elif what in ("bam", "fragments"):
    n = 200_000_000
    f_np, t_np, m_np, l_np = synth.make_bam_cores(2_000_000, seed=5)
    flag = torch.from_numpy(f_np.view(np.int16)).to(dev).repeat(100)
    tid, mtid, tlen = (torch.from_numpy(x).to(dev).repeat(100) for x in (t_np, m_np, l_np))
    if what == "bam":
        out = torch.zeros((4 + 5001,), dtype=torch.int64, device=dev)
        run = lambda: ctx.bam_flag_tlen_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 5000, out.data_ptr())
        bpu = 14
    else:
        bits = torch.empty(((n + 7) // 8,), dtype=torch.uint8, device=dev)
        kept = torch.zeros((1,), dtype=torch.int64, device=dev)
        run = lambda: ctx.bam_fragments_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 0, 5000, bits.data_ptr(), kept.data_ptr())
        bpu = 14.125
    units = n
elif what.startswith("sequence"):
    n, P = 16_000_000, int(what[8:])
    s4 = torch.randint(0, 256, (n, 76), dtype=torch.uint8, device=dev, generator=g)
    q = torch.randint(0, 42, (n, P), dtype=torch.uint8, device=dev, generator=g)
    ln = torch.full((n,), min(150, P), dtype=torch.int16, device=dev)
    fl = torch.randint(0, 2, (n,), dtype=torch.int16, device=dev, generator=g) * 16
    o = torch.empty((n, P), dtype=torch.uint8, device=dev)
    run = lambda: ctx.bam_sequence_dev(s4.data_ptr(), 76, q.data_ptr(), P, ln.data_ptr(), fl.data_ptr(), n, 10, o.data_ptr())
    units, bpu = n, 76 + 2 * P + 4
elif what.startswith("inflate") or what == "deflate":
    if what == "deflate":
        seq, qual = synth.make_reads(40000, 150, seed=31)
        text = synth.fastq_text(seq, qual, prefix="SIM:31") * 16
        B = 0xff00
        nb = -(-len(text) // B)
        blocks = np.zeros(nb, dtype=ctx.DEFLATE_BLOCK_DTYPE)
        for i in range(nb):
            blocks[i] = (i * B, min(B, len(text) - i * B), 0)
        src = np.frombuffer(text + bytes(8), dtype=np.uint8)
        d_in, d_blk = ctx.malloc_device(src.nbytes + 64), ctx.malloc_device(blocks.nbytes + 64)
        d_slots, d_tok = ctx.malloc_device(nb * 81920 + 64), ctx.malloc_device(nb * B * 4 + 64)
        d_res, d_crc = ctx.malloc_device(nb * 8 + 64), ctx.malloc_device(nb * 4 + 64)
        ctx.copy_h2d(d_in, src); ctx.copy_h2d(d_blk, blocks.view(np.uint8)); ctx.sync()
        run = lambda: ctx._check(ctx._lib.sk_bgzf_deflate_dev(ctx._h, d_in, d_blk, nb, d_slots, 81920, d_tok, d_res, d_crc), "sk_bgzf_deflate_dev")
        units, bpu, reps, warm = len(text), 1, 6, 6
    else:
        kind = "sorted" if what.endswith("sorted") else "random"
        path = f"/tmp/sk_kernel_one_{kind}.bam"
        synth.write_bam_file(path, 400_000, kind=kind, unit_records=100_000)
        data = open(path, "rb").read()
        os.remove(path)
        # the file's blocks, eight times over
        blks, at, out_off = [], 0, 0
        while at < len(data):
            xlen = struct.unpack_from("<H", data, at + 10)[0]
            bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
            crc, isize = struct.unpack_from("<II", data, at + bsize - 8)
            blks.append((at + 12 + xlen, bsize - 12 - xlen - 8, isize, crc))
            at += bsize
        R = 8
        blocks = np.zeros(len(blks) * R, dtype=ctx.BGZF_BLOCK_DTYPE)
        for r in range(R):
            for j, (io, il, ol, crc) in enumerate(blks):
                blocks[r * len(blks) + j] = (r * len(data) + io, il, ol, out_off, crc, 0)
                out_off += ol
        comp = np.frombuffer(data * R + bytes(64), dtype=np.uint8)
        d_comp, d_blocks = ctx.malloc_device(comp.nbytes + 64), ctx.malloc_device(blocks.nbytes + 64)
        d_out, d_status = ctx.malloc_device(out_off + 64), ctx.malloc_device(4 * len(blocks) + 64)
        ctx.copy_h2d(d_comp, comp); ctx.copy_h2d(d_blocks, blocks.view(np.uint8)); ctx.sync()
        run = lambda: ctx.bgzf_inflate_dev(d_comp, d_blocks, len(blocks), d_out, d_status, True)
        units, bpu, reps, warm = out_off, 1, 6, 6
else:
    raise SystemExit(__doc__)
torch.cuda.synchronize()
for _ in range(warm):
    run()
ctx.sync()
ctx.timer_start()
for _ in range(reps):
    run()
ms = ctx.timer_stop() / reps
print(f"{what}: {ms:.4f} ms per launch, {units / ms / 1e6:.2f} G units/s, {units * bpu / ms / 1e6:.1f} GB/s = {units * bpu / ms / 1e6 / 80:.1f} % of 8 TB/s")
