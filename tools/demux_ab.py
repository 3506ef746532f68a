#!/usr/bin/env python3
"""Demultiplex alone (decision only: assignment codes + counters) of several builds on the same barcode matrix in one
process, and of the lookup kernel's 1 / 4 tiles per wave iteration (DEMUX_TILES=1,4, or 1:0 for tiles:direct, 1:1:0 for tiles:direct:table in LDS; SK_DEMUX_TILES is read per launch).
usage: [SK_LIBS=tools/ab/x.so] [DEMUX_TILES=1,4] [DEMUX_N=10000000,100000000] python tools/demux_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
tiles = [x for x in os.environ.get("DEMUX_TILES", "").split(",") if x]
if tiles:
    ctxs = [(f"{name}/{nt}", ctx, nt) for name, ctx in ctxs for nt in tiles]
else:
    ctxs = [(name, ctx, None) for name, ctx in ctxs]
pad = int(os.environ.get("DEMUX_PAD", "0"))          # bytes added to every row: 17 -> 24 makes the dual-index rows start on dword boundaries, 8 -> 9 takes them off
sizes = [int(x) for x in os.environ.get("DEMUX_N", "10000000,100000000").split(",")]
for what, S, dual, L, bpu in (("16 single-index", 16, False, 8, 12), ("96 dual-index", 96, True, 17, 21)):
    table = synth.make_sheet(S, 8, dual=dual, seed=3 if not dual else 4)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3, halves=2 if dual else 1)
    for n in sizes:
        bc = torch.from_numpy(bc_np).to(dev).repeat(n // 1_000_000, 1)
        if pad:
            bc = torch.nn.functional.pad(bc, (0, pad), value=0x41)
        bc = bc.contiguous()
        assign = torch.empty((n,), dtype=torch.int32, device=dev)
        ref = None
        for name, ctx, nt in ctxs:
            if nt:
                os.environ["SK_DEMUX_TILES"] = nt.split(":")[0]
                if ":" in nt:
                    os.environ["SK_DEMUX_DIRECT"] = nt.split(":")[1]
                if nt.count(":") > 1:
                    os.environ["SK_DEMUX_LDSTAB"] = nt.split(":")[2]
            ctx.set_barcodes(table, 1)
            def run():
                ctx.demux_assign_dev(bc.data_ptr(), L + pad, n, assign.data_ptr())
            run(); ctx.sync()
            got = assign[:200000].clone()
            if ref is None:
                ref = got
            assert torch.equal(got, ref), name
            ts = []
            for _ in range(5):
                ctx.timer_start()
                for _ in range(10):
                    run()
                ts.append(ctx.timer_stop() / 10)
            ms = sorted(ts)[2]
            print(f"{what:16s} n={n:>10d} {name:8s}: {ms:7.4f} ms  {n / ms / 1e6:7.1f} G/s", flush=True)
        del bc, assign
