import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
for case, kw in (("exact", dict(p_exact=1.0, p_sub=0.0)), ("noisy", {})):
    b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
    bc = torch.from_numpy(b_np).to(dev).repeat(8, 1).contiguous()
    for ms in (1, 2, 4, 8, 16):
        os.environ["SK_CENSUS_MIN_STEPS"] = str(ms)
        ctx = seqkit_amd.Context(0)
        for n in (64_000, 250_000, 1_000_000, 4_000_000, 8_000_000):
            ts = []
            for _ in range(6):
                ctx.census_reset(); ctx.sync(); ctx.timer_start()
                ctx.census_add_dev(bc.data_ptr(), 17, 17, n, 0, 0)
                ts.append(ctx.timer_stop())
            print(case, "min_steps", ms, n, f"{sorted(ts)[2]:.3f}", flush=True)
