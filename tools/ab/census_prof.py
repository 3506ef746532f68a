import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
case = sys.argv[1]
kw = dict(p_exact=1.0, p_sub=0.0) if case == "exact" else {}
b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
bc = torch.from_numpy(b_np).to(dev).repeat(32, 1).contiguous()
ctx = seqkit_amd.Context(0, lib_path=os.path.abspath("tools/ab/prof.so"))
for _ in range(2):
    ctx.census_reset(); ctx.sync(); ctx.timer_start()
    ctx.census_add_dev(bc.data_ptr(), 17, 17, 32_000_000, 0, 0)
    print(case, f"{ctx.timer_stop():.3f} ms", flush=True)
