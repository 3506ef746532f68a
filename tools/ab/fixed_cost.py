import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
g = torch.Generator(device=dev); g.manual_seed(7)
N = 4_000_000
q = torch.randint(35, 74, (N, 150), dtype=torch.uint8, device=dev, generator=g)
lk = torch.empty((N,), dtype=torch.int16, device=dev)
table = synth.make_sheet(16, 8, dual=False, seed=3)
ctx.set_barcodes(table, 1)
bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=3)
bc = torch.from_numpy(bc_np).to(dev).repeat(40, 1).contiguous()
assign = torch.empty((40_000_000,), dtype=torch.int32, device=dev)
def t(fn, iters=20):
    for _ in range(3): fn()
    ctx.sync(); ts=[]
    for _ in range(5):
        ctx.timer_start()
        for _ in range(iters): fn()
        ts.append(ctx.timer_stop()/iters)
    return sorted(ts)[2]*1e3
for n in (64, 4096, 65536, 262144, 524288, 1_000_000, 2_000_000, 4_000_000):
    print(f"trim n={n:8d}: {t(lambda: ctx.trim_by_quality_dev(q.data_ptr(), 0, 150, n, 20, lk.data_ptr())):7.2f} us", flush=True)
for n in (64, 65536, 1_000_000, 2_500_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000):
    print(f"lut  n={n:8d}: {t(lambda: ctx.demux_assign_dev(bc.data_ptr(), 8, n, assign.data_ptr())):7.2f} us", flush=True)
