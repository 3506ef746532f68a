import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
libs = sorted(p for p in os.listdir("tools/ab") if p.startswith("comb_") and p.endswith(".so"))
ctxs = [(p, seqkit_amd.Context(0, lib_path=os.path.abspath("tools/ab/" + p))) for p in libs]
for case, kw in (("noisy", {}), ("clean", dict(p_exact=0.97, p_sub=0.025))):
    b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
    bc = torch.from_numpy(b_np).to(dev).repeat(32, 1).contiguous()
    for rnd in range(2):
        for name, ctx in ctxs:
            ts = []
            for _ in range(6):
                ctx.census_reset(); ctx.sync(); ctx.timer_start()
                ctx.census_add_dev(bc.data_ptr(), 17, 17, bc.shape[0], 0, 0)
                ts.append(ctx.timer_stop())
            print(case, name, " ".join(f"{t:.3f}" for t in sorted(ts[1:])), flush=True)
