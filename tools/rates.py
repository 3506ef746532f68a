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


# the rows of bench.py's extra.rates (one source for both): every BASELINE config device-resident; rows whose bytes would fit
# the Infinity Cache are timed with their rows coming from HBM (frac), pipelined and replayed on-die (frac_warm)
for r in bench.secondary_rates(torch, ctx, dev):
    extra = ""
    if "frac_warm" in r:
        extra = f"   [pipelined {r['ms_pipelined'] * 1e3:.1f} us {r['frac_pipelined']:.3f}; on-die replay {r['ms_warm'] * 1e3:.1f} us {r['frac_warm']:.3f}]"
    if "frac_as_placed" in r:
        extra += f"   [as placed {r['frac_as_placed']:.3f}]"
    print(f"{r['config'][:100]:100s} {r['ms'] * 1e3:9.1f} us {r['G_units_per_s']:8.2f} G units/s  {r['frac']:.3f} of 8 TB/s{extra}", flush=True)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)

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
