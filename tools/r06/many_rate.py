#!/usr/bin/env python3
"""Lookups of 10 M rows with their rows coming from HBM (cold buffer sets in rotation): one event pair per call, the calls back to
back inside one event pair, and the sets as the batches of ONE sk_demux_assign_many_dev call.  usage: many_rate.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
n = 10_000_000
for name, S, dual, L, bpu, detail in (("cfg3 16 x 8", 16, False, 8, 12, False), ("cfg3 16 x 8, detail of matched", 16, False, 8, 17, True),
                                       ("96 dual-index", 96, True, 17, 21, False), ("96 dual-index, detail of matched", 96, True, 17, 26, True),
                                       ("384 dual-index", 384, True, 17, 21, False)):
    table = synth.make_sheet(S, 8, dual=dual, seed=3 if not dual else (4 if S == 96 else 384))
    ctx.set_barcodes(table, 1)
    ctx.set_detail_mode(capi.SK_DETAIL_MATCHED if detail else capi.SK_DETAIL_FULL)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3, halves=2 if dual else 1)
    bc = torch.from_numpy(bc_np).to(dev).repeat(10, 1).contiguous()
    k = bench.cold_sets(n * bpu)
    keep, calls = [], []
    one_alloc = os.environ.get("MANY_ONE_ALLOC") == "1"                # (A/B: the sets as slices of ONE allocation per array)
    if one_alloc:
        big_b = bc.repeat(k, 1).contiguous()
        big_o = torch.empty((k * n,), dtype=torch.int32, device=dev)
    for i in range(k):
        b = (big_b[i * n:(i + 1) * n] if one_alloc else (bc if i == 0 else bc.clone()))
        o = [big_o[i * n:(i + 1) * n] if one_alloc else torch.empty((n,), dtype=torch.int32, device=dev)]
        if detail:
            o += [torch.empty((n,), dtype=torch.uint8, device=dev), torch.empty((n,), dtype=torch.int16, device=dev), torch.empty((n,), dtype=torch.int16, device=dev)]
        keep.append((b, o))
        calls.append(lambda b=b, o=o: ctx.demux_assign_dev(b.data_ptr(), L, n, *[x.data_ptr() for x in o]))
    many = lambda: ctx.demux_assign_many_dev([(b.data_ptr(), n, *[x.data_ptr() for x in o]) for b, o in keep], L)
    # warm the clocks (tools/r06/lut_repro.py)
    for _ in range(20):
        for f in calls:
            f()
    ctx.sync()
    ms = bench.measure_rotating(torch, ctx, dev, calls, rounds=5)
    stream = torch.cuda.ExternalStream(ctx.stream(), device=dev)

    def one_pair(fn, per):
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(stream):
                e0.record(stream)
                fn()
                e1.record(stream)
            ctx.sync()
            ts.append(e0.elapsed_time(e1) / per)
        return sorted(ts)[2]
    piped = one_pair(lambda: [f() for f in calls], k)
    many(); ctx.sync()
    m = one_pair(many, k)
    m3 = one_pair(lambda: [many() for _ in range(3)], 3 * k)        # three calls inside one event pair: the host's part of a call hides behind the launch before
    # the many-batch call's outputs equal the single calls'
    for b, o in keep[:2]:
        ref = torch.empty_like(o[0])
        ctx.demux_assign_dev(b.data_ptr(), L, n, ref.data_ptr()) if not detail else None
    frac = lambda t: n * bpu / t / 1e6 / 8000
    print(f"{name:36s} {k} sets: per call {ms * 1e3:6.1f} us = {frac(ms):.3f}; back to back {piped * 1e3:6.1f} us = {frac(piped):.3f}; ONE many-batch call {m * 1e3:6.1f} us = {frac(m):.3f}, three such calls in a row {m3 * 1e3:6.1f} us = {frac(m3):.3f}", flush=True)
    ctx.set_detail_mode(capi.SK_DETAIL_FULL)
    del keep, calls
