"""The census launch with environment knobs, same buffers, one process.  usage: python tools/ab/census_env_ab.py "K=V,K2=V2" ..."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
variants = sys.argv[1:] or [""]
import bench
for case, kw in (("noisy", {}), ("clean", dict(p_exact=0.97, p_sub=0.025)), ("noisy_indep", {}), ("clean_indep", dict(p_exact=0.97, p_sub=0.025))):
    if case.endswith("_indep"):                            # every row drawn independently instead of one million rows 32 times
        gi = torch.Generator(device=dev); gi.manual_seed(11)
        tt = torch.tensor(table, dtype=torch.uint8, device=dev)
        bases_t, alpha_t = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev), torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
        bc = torch.cat([bench.observed_barcodes(torch, gi, dev, 4_000_000, tt, bases_t, alpha_t, **kw) for _ in range(8)]).contiguous()
    else:
        b_np, _ = synth.observe_barcodes(table, 1_000_000, seed=4, halves=2, **kw)
        bc = torch.from_numpy(b_np).to(dev).repeat(32, 1).contiguous()
    for rnd in range(2):
        for v in variants:
            env = dict(e.split("=") for e in v.split(",") if e)
            os.environ.update(env)
            ts = []
            for _ in range(6):
                ctx.census_reset(); ctx.sync(); ctx.timer_start()
                ctx.census_add_dev(bc.data_ptr(), 17, 17, bc.shape[0], 0, 0)
                ts.append(ctx.timer_stop())
            for k in env: os.environ.pop(k, None)
            print(case, f"{v or 'default':30s}", " ".join(f"{t:.3f}" for t in sorted(ts[1:])), flush=True)
