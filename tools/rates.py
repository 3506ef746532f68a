#!/usr/bin/env python3
"""Device-resident rates of every BASELINE.json config on one GPU + the PCIe-inclusive rate of the host entry point.
usage: python tools/rates.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev)
g.manual_seed(7)


def timeit(name, fn, units, bytes_per_unit, iters=10, rounds=5):
    for _ in range(2):
        fn()
    ctx.sync()
    ts = []
    for _ in range(rounds):
        ctx.timer_start()
        for _ in range(iters):
            fn()
        ts.append(ctx.timer_stop() / iters)
    ms = sorted(ts)[len(ts) // 2]
    print(f"{name:58s} {ms:9.4f} ms  {units / ms / 1e3:10.1f} M units/s  {units * bytes_per_unit / ms / 1e6:8.1f} GB/s ({units * bytes_per_unit / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)


# cfg 2: trim by quality, 1M x 150 (and 16M for a bandwidth-sized run)
for n in (1_000_000, 16_000_000):
    q = torch.randint(35, 74, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    lk = torch.empty((n,), dtype=torch.int16, device=dev)
    timeit(f"cfg2 trim by quality {n} x 150bp (152 B/read)", lambda: ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, 20, lk.data_ptr()), n, 152)
    s = torch.randint(65, 85, (n, 150), dtype=torch.uint8, device=dev, generator=g)
    o = torch.empty_like(s)
    timeit(f"cfg1-shape mask by quality {n} x 150bp (450 B/read)", lambda: ctx.mask_by_quality_dev(s.data_ptr(), q.data_ptr(), 150, n, 20, o.data_ptr()), n, 450)
    if n == 16_000_000:
        # the fused pass in the forms BASELINE.md §3 names beside the paired headline (same rows as bench.py's extra.rates)
        from seqkit_amd import capi as _capi
        t16 = synth.make_sheet(16, 8, dual=False, seed=3)
        ctx.set_barcodes(t16, 1)
        b_np, _ = synth.observe_barcodes(t16, 1_000_000, seed=3)
        bc1 = torch.from_numpy(b_np).to(dev).repeat(16, 1).contiguous()
        assign = torch.empty((n,), dtype=torch.int32, device=dev)
        cnt = torch.zeros((96 + 3,), dtype=torch.int64, device=dev)
        n_mates = 1
        mate = {"seq": s.data_ptr(), "qual": q.data_ptr(), "len": 0, "out_seq": o.data_ptr(), "lowest_k": lk.data_ptr()}
        timeit("fused single-end 16M x 150bp + 8bp, 16 barcodes (464 B/read)",
               lambda: ctx.fused_pass_dev(n, 150, 20, [mate], bc=bc1.data_ptr(), bc_stride=8, assign=assign.data_ptr(), counts=cnt.data_ptr()), n, 464)
        ln = torch.randint(100, 151, (n,), dtype=torch.int16, device=dev, generator=g)
        mate_r = dict(mate, len=ln.data_ptr())
        timeit("fused single-end, ragged rows 100-150 (466 B/read)",
               lambda: ctx.fused_pass_dev(n, 150, 20, [mate_r], bc=bc1.data_ptr(), bc_stride=8, assign=assign.data_ptr(), counts=cnt.data_ptr()), n, 466)
        t96 = synth.make_sheet(96, 8, dual=True, seed=4)
        ctx.set_barcodes(t96, 1)
        b_np, _ = synth.observe_barcodes(t96, 1_000_000, seed=4, halves=2)
        bc2 = torch.from_numpy(b_np).to(dev).repeat(16, 1).contiguous()
        s2, q2, o2, lk2 = s.flip(0).contiguous(), q.flip(0).contiguous(), torch.empty_like(o), torch.empty_like(lk)
        low = torch.empty((n,), dtype=torch.uint8, device=dev)
        first = torch.empty((n,), dtype=torch.int16, device=dev)
        last = torch.empty((n,), dtype=torch.int16, device=dev)
        n_mates = 2
        mates2 = [mate, {"seq": s2.data_ptr(), "qual": q2.data_ptr(), "len": 0, "out_seq": o2.data_ptr(), "lowest_k": lk2.data_ptr()}]
        ctx.set_detail_mode(_capi.SK_DETAIL_MATCHED)
        timeit("fused paired + detail of matched clusters, 16M x 2x150bp, 96 dual-index (930 B/cluster)",
               lambda: ctx.fused_pass_dev(n, 150, 20, mates2, bc=bc2.data_ptr(), bc_stride=17, assign=assign.data_ptr(), lowest_diff=low.data_ptr(),
                                          first_idx=first.data_ptr(), last_idx=last.data_ptr(), counts=cnt.data_ptr()), n, 930)
        ctx.set_detail_mode(_capi.SK_DETAIL_FULL)
        del bc1, bc2, s2, q2, o2, lk2, low, first, last, assign, cnt, ln
    del q, lk, s, o

# cfg 3: demultiplex 10M x 8bp, 16 barcodes
n = 10_000_000
table = synth.make_sheet(16, 8, dual=False, seed=3)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3)
bc = torch.from_numpy(bc_np).to(dev).repeat(10, 1).contiguous()
assign = torch.empty((n,), dtype=torch.int32, device=dev)
timeit("cfg3 demultiplex 10M x 8bp, 16 barcodes (12 B/read)", lambda: ctx.demux_assign_dev(bc.data_ptr(), 8, n, assign.data_ptr()), n, 12)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
bc = torch.from_numpy(bc_np).to(dev).repeat(10, 1).contiguous()
timeit("demultiplex only 10M x 17ch, 96 dual-index (21 B/pair)", lambda: ctx.demux_assign_dev(bc.data_ptr(), 17, n, assign.data_ptr()), n, 21)
table = synth.make_sheet(384, 8, dual=True, seed=384)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2)
bc = torch.from_numpy(bc_np).to(dev).repeat(10, 1).contiguous()
timeit("demultiplex only 10M x 17ch, 384 dual-index, half by half (21 B/pair)", lambda: ctx.demux_assign_dev(bc.data_ptr(), 17, n, assign.data_ptr()), n, 21)
del bc, assign

# cfg 5: BAM flag + TLEN, 200M records
n = 200_000_000
flag_np, tid_np, mtid_np, tlen_np = synth.make_bam_cores(2_000_000, seed=5)
flag = torch.from_numpy(flag_np.view(np.int16)).to(dev).repeat(100)
tid = torch.from_numpy(tid_np).to(dev).repeat(100)
mtid = torch.from_numpy(mtid_np).to(dev).repeat(100)
tlen = torch.from_numpy(tlen_np).to(dev).repeat(100)
out = torch.zeros((4 + 5001,), dtype=torch.int64, device=dev)
timeit("cfg5 sam statistics + fragment lengths 200M records (14 B)", lambda: ctx.bam_flag_tlen_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 5000, out.data_ptr()), n, 14, iters=3, rounds=3)
bits = torch.zeros((n // 8 + 8,), dtype=torch.uint8, device=dev)
kept = torch.zeros((1,), dtype=torch.int64, device=dev)
timeit("f2 sam fragments filter 200M records (14 B)", lambda: ctx.bam_fragments_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 0, 5000, bits.data_ptr(), kept.data_ptr()), n, 14.125, iters=3, rounds=3)
del flag, tid, mtid, tlen, bits

# sam count: 100M coordinate-sorted paired records on 24 references, 20k regions of 1 kb (31 B/record)
n = 100_000_000
n_chr = 24
rng = np.random.default_rng(9)
ckey = torch.sort((torch.randint(0, n_chr, (n,), dtype=torch.int64, device=dev, generator=g) << 32) | torch.randint(0, 100_000_000, (n,), dtype=torch.int64, device=dev, generator=g)).values
ctid = (ckey >> 32).to(torch.int32).contiguous()          # sorted by (reference, position), as the host demands of its input
cpos = (ckey & 0xffffffff).to(torch.int32).contiguous()
del ckey
ctl = torch.randint(50, 600, (n,), dtype=torch.int32, device=dev, generator=g)
cflag = torch.full((n,), 99, dtype=torch.int16, device=dev)
cmapq = torch.full((n,), 60, dtype=torch.uint8, device=dev)
cmpos = (cpos + ctl // 2).contiguous()
rchr = np.sort(rng.integers(0, n_chr, size=20000)).astype(np.int32)
rstart = rng.integers(0, 100_000_000, size=20000).astype(np.uint32)
chr_off = np.searchsorted(rchr, np.arange(n_chr + 1)).astype(np.int32)
ctx.count_set_regions(chr_off, rstart, (rstart + 1000).astype(np.uint32))
timeit("sam count 100M records, 20k regions (31 B/record)", lambda: ctx.count_add_dev(cflag.data_ptr(), cmapq.data_ptr(), ctid.data_ptr(), ctid.data_ptr(), cpos.data_ptr(), cmpos.data_ptr(), ctl.data_ptr(), 0, n), n, 31, iters=3, rounds=3)
print("   regions hit:", int((ctx.count_get() > 0).sum()), "of 20000; fragments counted:", int(ctx.count_get().sum()) // (3 * 3 + 2))
del ctid, cpos, ctl, cflag, cmapq, cmpos

# fasta gc content: a 1 GB genome, 10 000 regions of 100 kb (1 B/base); host entry point, so the region list's upload,
# the launch and the read-back of 10 000 x 16 B are inside the time
genome = np.frombuffer(b"ACGTNacgtn", dtype=np.uint8)[rng.integers(0, 10, size=1_000_000_000)]
ctx.gc_set_genome(genome)
gs = rng.integers(0, 1_000_000_000 - 100_000, size=10000).astype(np.int64)
gl = np.full(10000, 100_000, dtype=np.int64)
ctx.gc_count(gs, gl)
t0 = time.perf_counter()
for _ in range(5):
    ctx.gc_count(gs, gl)
dt = (time.perf_counter() - t0) / 5
print(f"fasta gc content, 10 000 regions x 100 kb of a resident 1 GB genome: {dt * 1e3:.3f} ms  {1e9 / dt / 1e9:.1f} G bases/s = GB/s ({1e9 / dt / 8e12 * 100:.1f}% of 8 TB/s)", flush=True)
ctx.gc_set_genome(b"")
del genome

# f4: sam to fastq sequence(), 16M records x 152-base rows (0.5 + 1 + 1 B per base, + len and flag)
n = 16_000_000
s4 = torch.randint(0, 256, (n, 76), dtype=torch.uint8, device=dev, generator=g)
q = torch.randint(0, 42, (n, 152), dtype=torch.uint8, device=dev, generator=g)
ln = torch.full((n,), 150, dtype=torch.int16, device=dev)
fl = (torch.randint(0, 2, (n,), dtype=torch.int16, device=dev, generator=g) * 16).contiguous()
o = torch.empty_like(q)
timeit("f4 sam to fastq sequence() 16M x 150 bases (384 B/record)", lambda: ctx.bam_sequence_dev(s4.data_ptr(), 76, q.data_ptr(), 152, ln.data_ptr(), fl.data_ptr(), n, 10, o.data_ptr()), n, 76 + 152 + 152 + 4)
fl.zero_()
timeit("f4 sequence(), forward strand only", lambda: ctx.bam_sequence_dev(s4.data_ptr(), 76, q.data_ptr(), 152, ln.data_ptr(), fl.data_ptr(), n, 10, o.data_ptr()), n, 76 + 152 + 152 + 4)
del s4, q, ln, fl, o

# PCIe-inclusive: the host entry point sk_fused_pass (chunks of the batch in a two-lane pipeline: the H2D copies of chunk
# k+1 run under the kernel and the D2H copies of chunk k).  From pinned buffers (sk_malloc_pinned: what the C++ hosts pack
# into) the copies are DMA transfers; from pageable memory the runtime stages them.
n = 2_000_000
seq, qual, bcd = bench.gen_shard(torch, dev, n, table, seed=9, chunk=1_000_000)
import ctypes as C  # noqa: E402
from seqkit_amd import capi  # noqa: E402


def fused_host(arrs):
    """sk_fused_pass on caller-owned arrays (inputs and outputs), so that where they live is the caller's choice"""
    a = capi._FusedArgs()
    a.n, a.n_mates, a.stride, a.min_baseq = n, 2, 150, 20
    for i in range(2):
        a.mate[i].seq, a.mate[i].qual = arrs["seq"][i].ctypes.data, arrs["qual"][i].ctypes.data
        a.mate[i].out_seq, a.mate[i].lowest_k = arrs["out"][i].ctypes.data, arrs["lk"][i].ctypes.data
    a.bc, a.bc_stride, a.assign = arrs["bc"].ctypes.data, 17, arrs["assign"].ctypes.data
    ctx._check(ctx._lib.sk_fused_pass(ctx._h, C.byref(a)), "sk_fused_pass")


for kind in ("pinned", "pageable"):
    mk = (lambda shape, dt=np.uint8: ctx.pinned_empty(shape, dt)) if kind == "pinned" else (lambda shape, dt=np.uint8: np.empty(shape, dtype=dt))
    arrs = {"seq": [mk((n, 150)) for _ in range(2)], "qual": [mk((n, 150)) for _ in range(2)], "out": [mk((n, 150)) for _ in range(2)],
            "lk": [mk((n,), np.uint16) for _ in range(2)], "bc": mk((n, 17)), "assign": mk((n,), np.int32)}
    for i in range(2):
        arrs["seq"][i][:] = seq[i].cpu().numpy()
        arrs["qual"][i][:] = qual[i].cpu().numpy()
    arrs["bc"][:] = bcd.cpu().numpy()
    fused_host(arrs)
    t0 = time.perf_counter()
    for _ in range(3):
        fused_host(arrs)
    dt = (time.perf_counter() - t0) / 3
    print(f"host entry point sk_fused_pass, {n} pairs from {kind} host memory: {dt * 1e3:.1f} ms  {n / dt / 1e6:.2f} M pairs/s  {925 * n / dt / 1e9:.2f} GB/s (PCIe-inclusive)", flush=True)
    if kind == "pinned":
        ref = [arrs["out"][0][:1000].copy(), arrs["lk"][1][:1000].copy(), arrs["assign"][:1000].copy()]
    else:
        assert np.array_equal(ref[0], arrs["out"][0][:1000]) and np.array_equal(ref[1], arrs["lk"][1][:1000]) and np.array_equal(ref[2], arrs["assign"][:1000])
