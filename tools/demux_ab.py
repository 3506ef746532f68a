#!/usr/bin/env python3
"""Demultiplex alone of several builds and forms of the lookup kernel on the same barcode matrix in one process.
usage: [SK_LIBS=tools/ab/x.so,...] [DEMUX_FORMS="default;SK_DEMUX_LDSTAB=0;SK_NO_HASH_DEMUX=1"] [DEMUX_N=10000000,100000000]
       [DEMUX_PAD=7] [DEMUX_DETAIL=1] python tools/demux_ab.py
DEMUX_DETAIL=1 also times the SK_DETAIL_MATCHED form (lowest_diff / first / last of matched rows written too)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
forms = [f for f in os.environ.get("DEMUX_FORMS", "default").split(";") if f]
KNOBS = ("SK_DEMUX_LDSTAB", "SK_DEMUX_DIRECT", "SK_NO_HASH_DEMUX", "SK_DEMUX_ROWS2")
pad = int(os.environ.get("DEMUX_PAD", "0"))          # bytes added to every row: 17 -> 24 puts the dual-index rows on dword boundaries, 8 -> 9 takes them off
sizes = [int(x) for x in os.environ.get("DEMUX_N", "10000000,100000000").split(",")]
details = [False, True] if os.environ.get("DEMUX_DETAIL") else [False]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
for what, S, dual, L, bpu in (("16 single-index", 16, False, 8, 12), ("96 dual-index", 96, True, 17, 21)):
    table = synth.make_sheet(S, 8, dual=dual, seed=3 if not dual else 4)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3 if not dual else 4, halves=2 if dual else 1)
    for n in sizes:
        bc = torch.from_numpy(bc_np).to(dev).repeat(max(1, n // 1_000_000), 1)[:n]
        if pad:
            bc = torch.nn.functional.pad(bc, (0, pad), value=0x41)
        bc = bc.contiguous()
        assign = torch.empty((n,), dtype=torch.int32, device=dev)
        low = torch.empty((n,), dtype=torch.uint8, device=dev)
        first = torch.empty((n,), dtype=torch.int16, device=dev)
        last = torch.empty((n,), dtype=torch.int16, device=dev)
        ref = None
        for name, ctx in ctxs:
            for form in forms:
                for k in KNOBS:
                    os.environ.pop(k, None)
                if form != "default":
                    for kv in form.split(","):
                        k, v = kv.split("=")
                        os.environ[k] = v
                for detail in details:
                    ctx.set_barcodes(table, 1)
                    if detail:
                        ctx.set_detail_mode(seqkit_amd.SK_DETAIL_MATCHED)

                    def run():
                        if detail:
                            ctx.demux_assign_dev(bc.data_ptr(), L + pad, n, assign.data_ptr(), low.data_ptr(), first.data_ptr(), last.data_ptr())
                        else:
                            ctx.demux_assign_dev(bc.data_ptr(), L + pad, n, assign.data_ptr())
                    run(); ctx.sync()
                    got = assign[:min(n, 200000)].clone()
                    if ref is None:
                        ref = got
                    assert torch.equal(got, ref), (name, form)
                    ts = []
                    for _ in range(5):
                        ctx.timer_start()
                        for _ in range(10):
                            run()
                        ts.append(ctx.timer_stop() / 10)
                    ms = sorted(ts)[2]
                    b = bpu + (5 if detail else 0)
                    print(f"{what:16s} n={n:>10d} {name:8s} {form:24s} {'detail' if detail else 'decision'}: {ms:7.4f} ms  {n / ms / 1e6:7.1f} G/s  "
                          f"{n * b / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
                    if detail:
                        ctx.set_detail_mode(seqkit_amd.SK_DETAIL_FULL)
        del bc, assign, low, first, last
