import os, sys
sys.path.insert(0, os.getcwd())
import torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
n = int(os.environ.get("N", 10_000_000))
libs = sys.argv[1:]
for S, dual, seed in ((16, False, 3), (96, True, 4), (384, True, 384)):
    table = synth.make_sheet(S, 8, dual=dual, seed=seed)
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=seed, halves=2 if dual else 1)
    bc = torch.from_numpy(bc_np).to(dev).repeat(max(1, n // 1_000_000), 1)[:n].contiguous()
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    for name in libs * 2:
        ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(f"tools/ab/{name}.so"))
        ctx.set_barcodes(table, 1)
        for _ in range(3): ctx.demux_assign_dev(bc.data_ptr(), bc.shape[1], n, assign.data_ptr())
        ctx.sync(); ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10): ctx.demux_assign_dev(bc.data_ptr(), bc.shape[1], n, assign.data_ptr())
            ts.append(ctx.timer_stop() / 10)
        print(S, n, name, f"{sorted(ts)[2] * 1e3:.2f} us", flush=True)
        ctx.close()
