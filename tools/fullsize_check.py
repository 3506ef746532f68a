#!/usr/bin/env python3
"""BASELINE.json's configurations at their FULL sizes on one GPU, checked through the C-ABI: bit-exact against the oracle where
the oracle finishes in seconds (cfg 2, cfg 3, cfg 5 by construction), and through size-independent properties where it does not
(cfg 4's per-GPU shard of 62.5 M clusters: tiling independence, idempotence of the mask, counter identities, an exact
oracle comparison of a sample).  Run by tests/test_gpu_fullsize.py in its own process (torch must load its HIP runtime first).
The barcode census (row f3) rides along at a launch size that takes its partition path by itself.
usage: python tools/fullsize_check.py [cfg2] [cfg3] [cfg4] [cfg5] [census]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from oracle import oracle  # noqa: E402
from seqkit_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
which = set(sys.argv[1:]) or {"cfg2", "cfg3", "cfg4", "cfg5", "census"}


def to_dev(a):
    return torch.from_numpy(a).to(dev)


def cfg2():
    """fasta trim by quality, 1 M x 150 bp: bit-exact, every threshold class."""
    n = 1_000_000
    _, qual = synth.make_reads(n, 150, seed=2)
    qual = synth.add_forced_classes(qual, seed=2)
    q = to_dev(qual)
    lk = torch.empty((n,), dtype=torch.int16, device=dev)
    for m in (20, 0, 41):
        ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, m, lk.data_ptr())
        ctx.sync()
        got = lk.cpu().numpy().view(np.uint16)
        assert np.array_equal(got, oracle.trim_batch(qual, None, m)), m
    print("cfg2 ok: 1 M x 150 bp trim by quality == oracle at min_baseq 20, 0, 41", flush=True)


def cfg3():
    """fasta demultiplex, 10 M reads, 16 single-index 8 bp barcodes, <= 1 mismatch: bit-exact codes and counters."""
    n = 10_000_000
    table = synth.make_sheet(16, 8, seed=3)
    bc, _ = synth.observe_barcodes(table, n, seed=3)
    ctx.set_barcodes(table, 1)
    ctx.counts_reset()
    d = to_dev(bc)
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    ctx.demux_assign_dev(d.data_ptr(), 8, n, assign.data_ptr())
    ctx.sync()
    e_assign, _, _, _, e_counts = oracle.demux_batch(table, bc, 1)
    assert np.array_equal(assign.cpu().numpy(), e_assign)
    counts = ctx.counts()
    assert np.array_equal(counts, e_counts) and int(counts[:16].sum()) == int(counts[17]) and int(counts[16]) == n
    # what `fasta demultiplex` asks for: the same lookup with lowest_diff / first / last of the reads that matched something
    e_assign, e_low, e_first, e_last, _ = oracle.demux_batch(table, bc, 1)
    low = torch.empty((n,), dtype=torch.uint8, device=dev)
    first = torch.empty((n,), dtype=torch.int16, device=dev)
    last = torch.empty((n,), dtype=torch.int16, device=dev)
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
    ctx.demux_assign_dev(d.data_ptr(), 8, n, assign.data_ptr(), low.data_ptr(), first.data_ptr(), last.data_ptr())
    ctx.sync()
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_FULL)
    m = e_assign != -1
    assert np.array_equal(assign.cpu().numpy(), e_assign)
    assert np.array_equal(low.cpu().numpy()[m], e_low[m]) and np.array_equal(first.cpu().numpy()[m], e_first[m]) and np.array_equal(last.cpu().numpy()[m], e_last[m])
    assert np.array_equal(ctx.counts(), 2 * e_counts)
    print(f"cfg3 ok: 10 M x 8 bp, 16 barcodes == oracle (decision; decision + detail of the {int(m.sum())} matched reads); "
          f"identified {int(counts[17])}, ambiguous {int(counts[18])}", flush=True)
    # the metric's own sheet, demultiplex alone: 10 M x 17 ch, 96 dual-index
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=4, halves=2)
    bc[::1009, 8] = ord("-")                                    # a broken separator now and then
    ctx.set_barcodes(table, 1)
    d = to_dev(bc)
    e_assign, e_low, e_first, e_last, e_counts = oracle.demux_batch(table, bc, 1)
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)
    ctx.demux_assign_dev(d.data_ptr(), 17, n, assign.data_ptr(), low.data_ptr(), first.data_ptr(), last.data_ptr())
    ctx.sync()
    ctx.set_detail_mode(seqkit_amd.SK_DETAIL_FULL)
    m = e_assign != -1
    assert np.array_equal(assign.cpu().numpy(), e_assign) and np.array_equal(ctx.counts(), e_counts)
    assert np.array_equal(low.cpu().numpy()[m], e_low[m]) and np.array_equal(first.cpu().numpy()[m], e_first[m]) and np.array_equal(last.cpu().numpy()[m], e_last[m])
    print(f"96 dual-index ok: 10 M x 17 ch == oracle (decision + detail of the {int(m.sum())} matched pairs)", flush=True)


def cfg4():
    """fasta add barcode + demultiplex fused with trim + mask, one GPU's shard of configs[3] (62.5 M clusters)."""
    n, L, LB = 62_500_000, 150, 17
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    ctx.set_barcodes(table, 1)
    seq, qual, bc = bench.gen_shard(torch, dev, n, table, seed=4000, chunk=2_000_000)
    out = [torch.empty_like(seq[0]) for _ in range(2)]
    lk = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(2)]
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    counts = torch.zeros((99,), dtype=torch.int64, device=dev)

    def mates(lo, cnt_rows, src=seq, dst=out, k=lk):
        return [{"seq": src[i].data_ptr() + lo * L, "qual": qual[i].data_ptr() + lo * L, "len": 0, "out_seq": dst[i].data_ptr() + lo * L,
                 "lowest_k": k[i].data_ptr() + lo * 2} for i in range(2)]

    ctx.fused_pass_dev(n, L, 20, mates(0, n), bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr(), counts=counts.data_ptr())
    ctx.sync()
    c = counts.cpu().numpy()
    # counter identities (src/fasta_demultiplex.rs:108-109,169,176-178)
    assert int(c[96]) == n and int(c[:96].sum()) == int(c[97]) and int(c[97]) + int(c[98]) <= n
    assert int((assign >= 0).sum()) == int(c[97]) and int((assign == -2).sum()) == int(c[98])
    hist = torch.bincount(assign[assign >= 0], minlength=96).cpu().numpy()
    assert np.array_equal(hist, c[:96])
    # mask: an output byte is 'N' exactly where the quality is below the threshold, else the input byte
    for i in range(2):
        low = (qual[i] - 33) < 20                                # uint8 arithmetic wraps like the reference's
        assert bool(((out[i] == ord("N")) | ~low).all()) and bool(((out[i] == seq[i]) | low).all())
        del low
    # idempotence: masking the masked bases changes nothing (checksum of every row block)
    out2 = [torch.empty_like(seq[0]) for _ in range(2)]
    lk2 = [torch.empty_like(lk[0]) for _ in range(2)]
    ctx.fused_pass_dev(n, L, 20, mates(0, n, src=out, dst=out2, k=lk2))
    ctx.sync()
    for i in range(2):
        assert torch.equal(out2[i], out[i]) and torch.equal(lk2[i], lk[i])
    # tiling independence: the shard in five uneven launches (cut at multiples of 16 rows, the C-ABI's 16-byte alignment of
    # every matrix, but not of 64) gives the same bytes as in one
    cuts = [0, 64_000, 7_064_016, 30_000_000, 62_499_984, n]
    a3 = torch.empty_like(assign)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ctx.fused_pass_dev(hi - lo, L, 20, mates(lo, hi - lo, dst=out2, k=lk2), bc=bc.data_ptr() + lo * LB, bc_stride=LB, assign=a3.data_ptr() + lo * 4)
    ctx.sync()
    assert torch.equal(a3, assign)
    for i in range(2):
        assert torch.equal(out2[i], out[i]) and torch.equal(lk2[i], lk[i])
    # the tile-blocked layout (what bench.py times): the whole shard repacked into blocks gives the same bytes and counters
    # as the row-major matrices, on all 62.5 M clusters (the last tile is partial: 62.5 M = 976 562 x 64 + 32)
    del out2, lk2, a3
    torch.cuda.empty_cache()
    from seqkit_amd import capi
    nt = (n + 63) // 64
    pad = nt * 64 - n
    lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
    z = lambda t, w: torch.cat([t, torch.zeros((pad, w), dtype=torch.uint8, device=dev)]) if pad else t
    bin_, bout = bench.pack_blocked(torch, lay, [z(x, L) for x in seq], [z(x, L) for x in qual], z(bc, LB), nt)
    counts.zero_()
    torch.cuda.synchronize()
    ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())
    ctx.sync()
    assert np.array_equal(counts.cpu().numpy(), c)
    u = bench.unpack_blocked(torch, lay, bout, nt)
    assert torch.equal(u["assign"][:n], assign)
    for i in range(2):
        assert torch.equal(u["out_seq"][i][:n], out[i]) and torch.equal(u["lowest_k"][i][:n], lk[i])
    del u, bin_, bout
    # trim: 0 <= lowest_k <= 150, and rows whose last base has quality > 70 above the threshold break at once
    for i in range(2):
        k = lk[i].to(torch.int32)
        assert int(k.min()) >= 0 and int(k.max()) <= L
    # an exact sample from the middle and the end of the shard
    for lo, m in ((31_250_000, 500_000), (n - 300_001, 300_001)):
        hb = bc[lo:lo + m].cpu().numpy()
        e_assign = oracle.demux_batch(table, hb, 1)[0]
        assert np.array_equal(assign[lo:lo + m].cpu().numpy(), e_assign)
        for i in range(2):
            hq, hs = qual[i][lo:lo + m].cpu().numpy(), seq[i][lo:lo + m].cpu().numpy()
            assert np.array_equal(lk[i][lo:lo + m].cpu().numpy().view(np.uint16), oracle.trim_batch(hq, None, 20))
            assert np.array_equal(out[i][lo:lo + m].cpu().numpy(), oracle.mask_batch(hs, hq, None, 20))
    print(f"cfg4 ok: 62.5 M clusters x 2x150 bp: counters consistent ({int(c[97])} identified, {int(c[98])} ambiguous), mask rule holds "
          "everywhere, idempotent, tiling-independent, tile-blocked layout == row-major matrices, 800 k clusters == oracle", flush=True)


def cfg5():
    """sam statistics + sam fragment lengths, 200 M records: 100 repeats of a 2 M-record unit, so the oracle's answer for the
    unit times 100 is the exact expectation."""
    unit, reps = 2_000_000, 100
    flag, tid, mtid, tlen = synth.make_bam_cores(unit, seed=5)
    n = unit * reps
    d = [to_dev(x.view(np.int16) if x.dtype == np.uint16 else x).repeat(reps) for x in (flag, tid, mtid, tlen)]
    outv = torch.zeros((4 + 5001,), dtype=torch.int64, device=dev)
    ctx.bam_flag_tlen_dev(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), n, 5000, outv.data_ptr())
    ctx.sync()
    got = outv.cpu().numpy()
    counters, hist, total = oracle.bam_flag_tlen(flag, tid, mtid, tlen, 5000)
    assert [int(x) for x in got[:3]] == [int(x) * reps for x in counters] and int(got[3]) == total * reps
    assert np.array_equal(got[4:], hist.astype(np.int64) * reps) and int(got[0]) <= n          # secondary / supplementary records are not counted
    print(f"cfg5 ok: 200 M records: total {int(got[0])}, aligned {int(got[1])}, duplicate {int(got[2])}, histogram of {int(got[3])} fragments == 100 x oracle(unit)",
          flush=True)


def census():
    """f3 at 32 M rows in one launch (large launches write the keys their front tables have no room for out, partition and
    combine them): 32 repeats of a 1 M-row noisy dual-index unit, so every count is 32 x the oracle's for the unit and every
    first row is the unit's; then 16 M rows of which more than half are new keys (those are inserted as they lie)."""
    unit, reps = 1_000_000, 32
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, unit, seed=4, halves=2)
    d = to_dev(bc).repeat(reps, 1).contiguous()
    ctx.census_reset()
    ctx.census_add_dev(d.data_ptr(), 17, 17, unit * reps, 0, 0)
    want = oracle.census(bc, L=17)
    got, total = ctx.census_entries()
    st = ctx.census_stats()
    assert total == len(want) and got == [(b, c * reps, f) for b, c, f in want] and st["counted"] == unit * reps and st["rejected"] == 0
    # the dry-run shape at a row pitch of 24 (three tiles per wave step): only the rows demultiplexing left unassigned
    ctx.set_barcodes(table, 1)
    unit_assign = oracle.demux_batch(table, bc, 1)[0]
    wide = np.zeros((unit, 24), dtype=np.uint8)
    wide[:, :17] = bc
    wide[:, 17:] = np.frombuffer(b"#", dtype=np.uint8)                         # what lies beyond L is not looked at
    dw = to_dev(wide).repeat(reps, 1).contiguous()
    da = to_dev(unit_assign.astype(np.int32)).repeat(reps).contiguous()
    ctx.census_reset()
    ctx.census_add_dev(dw.data_ptr(), 24, 17, unit * reps, da.data_ptr(), 1000)
    want = oracle.census(bc, L=17, assign=unit_assign)
    got, total = ctx.census_entries()
    assert total == len(want) and got == [(b, c * reps, f + 1000) for b, c, f in want] and ctx.census_stats()["counted"] == reps * int((unit_assign == -1).sum())
    del dw, da
    n2 = 16_000_000
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    codes = torch.randint(0, 4, (n2, 12), device=dev, generator=g, dtype=torch.int64)      # 16.7 M possible 12-mers: most rows new, many twice
    rnd = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[codes].contiguous()
    ctx.census_reset()
    ctx.census_add_dev(rnd.data_ptr(), 12, 12, n2, 0, 0)
    vals = (codes << (2 * torch.arange(12, device=dev))).sum(dim=1)
    uniq, counts = torch.unique(vals, return_counts=True)
    st = ctx.census_stats()
    assert st["distinct"] == int(uniq.numel()) and st["counted"] == n2
    hist = ctx.census_count_hist()
    for b in range(6):
        assert int(hist[b]) == int(((counts >= (1 << b)) & (counts < (2 << b))).sum()), b
    top, _ = ctx.census_entries(min_count=int(counts.max()))
    assert len(top) == int((counts == counts.max()).sum()) and all(c == int(counts.max()) for _, c, _ in top)
    print(f"census ok: 32 M noisy rows == 32 x oracle(unit), every row and (pitch 24, row_base 1000) the unassigned ones ({len(want)} barcodes); 16 M random 12-mers: {int(uniq.numel())} distinct, count histogram and top entries as torch.unique has them",
          flush=True)


t0 = time.time()
for name, fn in (("cfg2", cfg2), ("cfg3", cfg3), ("cfg5", cfg5), ("census", census), ("cfg4", cfg4)):
    if name in which:
        fn()
        torch.cuda.empty_cache()
print(f"fullsize ok in {time.time() - t0:.1f} s")
