import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
n = 32_000_000
bc = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[torch.randint(0, 4, (n, 16), device=dev, generator=g)].contiguous()
for name, path in (("product", None), ("no first peek", os.path.abspath("tools/ab/exp_nopeek.so"))) * 2:
    ctx = seqkit_amd.Context(0, lib_path=path)
    ts = []
    for _ in range(4):
        ctx.census_reset(); ctx.sync(); ctx.timer_start()
        ctx.census_add_dev(bc.data_ptr(), 16, 16, n, 0, 0)
        ts.append(ctx.timer_stop())
    print(name, " ".join(f"{t:.3f}" for t in ts), ctx.census_stats()["distinct"], flush=True)
    ctx.close()
