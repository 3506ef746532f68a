#!/usr/bin/env python3
"""Where do the waves of the census front kernel spend their cycles?  Needs the diagnostic build (s_memtime stamps per phase):
   (cd seqkit_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSK_CENSUS_STAMPS -o ../../tools/ab/census_stamps.so \
       sk_kernels.hip sk_census.hip sk_capi.hip sk_lut.cpp -ldl)
usage: python tools/census_stamps.py [rows]  ->  per case: mean shader cycles per wave and phase, share of the wave's lifetime"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

LIB = os.path.abspath(os.environ.get("SK_STAMPS_LIB", "tools/ab/census_stamps.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0, lib_path=LIB)
lib = C.CDLL(LIB)
PHASES = ["tables cleared", "step's loads waited for, rows aligned", "fence + next step's loads issued", "rows the front table knows", "long way", "parked keys: records / inserts",
          "loop exit", "end barrier", "front table leaves the workgroup", "statistics"]
table = synth.make_sheet(96, 8, dual=True, seed=4)
CASES = os.environ.get("SK_STAMPS_CASES", "exact,clean,noisy").split(",")
for case, kw in (("exact", dict(p_exact=1.0, p_sub=0.0)), ("clean", dict(p_exact=0.97, p_sub=0.025)), ("noisy", {}), ("noisy_indep", {})):
    if case not in CASES:
        continue
    if case.endswith("_indep"):                                # every row drawn independently (bench.observed_barcodes), not one million rows repeated
        import bench
        gi = torch.Generator(device=dev)
        gi.manual_seed(11)
        tt = torch.tensor(table, dtype=torch.uint8, device=dev)
        bases_t, alpha_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev), torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
        bc = torch.cat([bench.observed_barcodes(torch, gi, dev, 4_000_000, tt, bases_t, alpha_t, **kw) for _ in range(max(1, n // 4_000_000))]).contiguous()
    else:
        b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
        bc = torch.from_numpy(b_np).to(dev).repeat(max(1, n // 1_000_000), 1).contiguous()
    for alias in ("-",):
        for _ in range(2):
            ctx.census_reset(); ctx.sync(); ctx.timer_start()
            ctx.census_add_dev(bc.data_ptr(), 17, 17, bc.shape[0], 0, 0)
            ms = ctx.timer_stop()
        waves = 256 * 16
        buf = np.zeros((waves, 16), dtype=np.uint64)
        rc = lib.sk_debug_census_stamps(buf.ctypes.data_as(C.c_void_p), waves)
        tot = buf[:, :10].sum(axis=1).astype(np.float64)
        print(f"== {case}: launch sequence {ms:.3f} ms; wave lifetime mean {tot.mean():.0f} cycles (min {tot.min():.0f} max {tot.max():.0f}) rc {rc}")
        for i, name in enumerate(PHASES):
            c = buf[:, i].astype(np.float64)
            print(f"   {name:28s} {c.mean():10.0f} cycles  {100 * c.mean() / tot.mean():5.1f} %   (min {c.min():.0f} max {c.max():.0f})")
        cw = 2 * 256 * 8
        cbuf = np.zeros((cw, 16), dtype=np.uint64)
        if hasattr(lib, "sk_debug_combine_stamps") and lib.sk_debug_combine_stamps(cbuf.ctypes.data_as(C.c_void_p), cw) == 0:
            ctot = cbuf[:, :9].sum(axis=1).astype(np.float64)
            print(f"   -- combine kernel, mean cycles per wave {ctot.mean():.0f}")
            for i, name in enumerate(["statistics, records counted, items scanned", "an item taken, the table cleared", "the bucket's regions scanned", "next records asked for, keys hashed",
                                      "counted in the table", "keys without a place: to HBM", "barrier behind the records", "occupied slots listed and inserted", "end"]):
                c = cbuf[:, i].astype(np.float64)
                print(f"      {name:32s} {c.mean():10.0f} cycles  {100 * c.mean() / max(ctot.mean(), 1):5.1f} %")
        sys.stdout.flush()
