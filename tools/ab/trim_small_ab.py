"""cfg 2 at its own size (1 M x 150) with its rows coming from HBM, under launch-shape knobs.  usage: python tools/ab/trim_small_ab.py "K=V" ..."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench, seqkit_amd
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev); g.manual_seed(7)
n = 1_000_000
k = bench.cold_sets(n * 152)
q = torch.randint(35, 74, (k * n, 150), dtype=torch.uint8, device=dev, generator=g)
lk = torch.empty((k * n,), dtype=torch.int16, device=dev)
for rnd in range(2):
    for v in (sys.argv[1:] or [""]):
        env = dict(e.split("=") for e in v.split(",") if e)
        os.environ.update(env)
        calls = [(lambda i=i: ctx.trim_by_quality_dev(q[i * n:].data_ptr(), 0, 150, n, 20, lk[i * n:].data_ptr())) for i in range(k)]
        cold = bench.measure_rotating(torch, ctx, dev, calls, rounds=6)
        warm = bench.measure_rotating(torch, ctx, dev, [calls[0]] * 8, rounds=3)
        for kk in env: os.environ.pop(kk, None)
        print(f"{v or 'default':24s} cold {cold * 1e3:6.2f} us {n * 152 / cold / 8e9:.3f}   warm {warm * 1e3:6.2f} us {n * 152 / warm / 8e9:.3f}", flush=True)
