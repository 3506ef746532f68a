#!/usr/bin/env python3
"""A/B of the lookup kernels with their rows coming from HBM (rotating buffer sets, bench.measure_rotating) and replayed on
one set (Infinity-Cache-warm), at 10 M rows and in one 100 M-row call.
usage: python tools/lut_cold_ab.py [--sheets 16,96,384] [--n 10000000,100000000] lib[:ENV=V,...] ...
  lib = a library under tools/ab/ (name without .so) or `cur` for seqkit_amd/lib/libseqkit_hip.so"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
args = sys.argv[1:]
sheets, ns, detail = [16, 96, 384], [10_000_000, 100_000_000], False
while args and args[0].startswith("--"):
    k = args.pop(0)
    if k == "--sheets":
        sheets = [int(x) for x in args.pop(0).split(",")]
    elif k == "--n":
        ns = [int(x) for x in args.pop(0).split(",")]
    elif k == "--detail":
        detail = True
variants = args or ["cur"]
SHEET = {16: (False, 3), 96: (True, 4), 384: (True, 384), 1000: (True, 1000)}
for S in sheets:
    dual, seed = SHEET[S]
    table = synth.make_sheet(S, 8, dual=dual, seed=seed)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=seed, halves=2 if dual else 1)
    L = bc_np.shape[1]
    for n in ns:
        bpu = L + 4 + (5 if detail else 0)
        k = bench.cold_sets(n * bpu)
        base = torch.from_numpy(bc_np).to(dev).repeat(max(1, n // 1_000_000), 1)[:n].contiguous()
        bcs = [base] + [base.clone() for _ in range(k - 1)]
        outs = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(k)]
        low = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(k)] if detail else None
        fi = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(k)] if detail else None
        la = [torch.empty((n,), dtype=torch.int16, device=dev) for _ in range(k)] if detail else None
        for v in variants * 2:
            name, _, envs = v.partition(":")
            env = dict(e.split("=") for e in envs.split(",") if e)
            for kk, vv in env.items():
                os.environ[kk] = vv
            path = seqkit_amd.library_path() if name == "cur" else os.path.abspath(f"tools/ab/{name}.so")
            ctx = seqkit_amd.Context(0, lib_path=path)
            ctx.set_barcodes(table, 1)
            if detail:
                ctx.set_detail_mode(capi.SK_DETAIL_MATCHED)

            def call(i):
                if detail:
                    return lambda: ctx.demux_assign_dev(bcs[i].data_ptr(), L, n, outs[i].data_ptr(), low[i].data_ptr(), fi[i].data_ptr(), la[i].data_ptr())
                return lambda: ctx.demux_assign_dev(bcs[i].data_ptr(), L, n, outs[i].data_ptr())
            cold = bench.measure_rotating(torch, ctx, dev, [call(i) for i in range(k)], rounds=6)
            warm = bench.measure_rotating(torch, ctx, dev, [call(0)] * 10, rounds=3)
            gb = n * bpu / 1e6
            print(f"S={S:4d} n={n:9d} sets={k} {v:40s} cold {cold * 1e3:8.2f} us {gb / cold / 8000:.3f}   warm {warm * 1e3:8.2f} us {gb / warm / 8000:.3f}", flush=True)
            ctx.close()
            for kk in env:
                os.environ.pop(kk, None)
        del bcs, outs, base, low, fi, la
