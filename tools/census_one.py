#!/usr/bin/env python3
"""One shape of the census, a few launches (for rocprofv3 runs).  usage: census_one.py exact|clean|noisy|sub|distinct|noisy_indep|clean_indep [rows] [reps]
(*_indep: every row drawn independently on the device — bench.observed_barcodes — instead of one million drawn rows repeated)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "exact"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32_000_000
reps_run = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(os.environ["SK_LIB"])) if os.environ.get("SK_LIB") else seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
kw = {"exact": dict(p_exact=1.0, p_sub=0.0), "clean": dict(p_exact=0.97, p_sub=0.025), "noisy": {}, "sub": dict(p_exact=0.85, p_sub=0.15)}
reps = max(1, n // 1_000_000)
if case.endswith("_indep"):
    import bench
    gi = torch.Generator(device=dev)
    gi.manual_seed(11)
    tt = torch.tensor(table, dtype=torch.uint8, device=dev)
    bases_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    alpha_t = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    bc = torch.cat([bench.observed_barcodes(torch, gi, dev, min(4_000_000, n), tt, bases_t, alpha_t, **kw[case[:-6]]) for _ in range(max(1, n // 4_000_000))]).contiguous()
    L = 17
elif case == "distinct":
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    bc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, 16), device=dev, generator=g)].contiguous()
    L = 16
else:
    b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw[case])
    bc = torch.from_numpy(b_np).to(dev).repeat(reps, 1).contiguous()
    L = 17
ts, rs = [], []
for _ in range(reps_run):
    ctx.sync()
    ctx.timer_start()
    ctx.census_reset()
    rs.append(ctx.timer_stop())
    ctx.timer_start()
    ctx.census_add_dev(bc.data_ptr(), bc.shape[1], L, bc.shape[0], 0, 0)
    ts.append(ctx.timer_stop())
st = ctx.census_stats()
print(f"{case} n={bc.shape[0]}: " + " ".join(f"{t:.3f}" for t in ts) + " ms  reset " + " ".join(f"{t:.3f}" for t in rs) + f" ms  distinct {st['distinct']} counted {st['counted']}")
