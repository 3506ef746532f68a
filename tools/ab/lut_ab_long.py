import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, seqkit_amd
from seqkit_amd import synth
dev = torch.device("cuda", 0)
n = 10_000_000
libs = sys.argv[1:]
rng = np.random.default_rng(7)
for S, L, stride in ((48, 12, 12), (48, 12, 13), (24, 20, 20), (24, 17, 24), (16, 8, 9)):
    table = np.unique(synth.BASES[rng.integers(0, 4, size=(4 * S, L))], axis=0)[:S].copy()
    bc_np, _ = synth.observe_barcodes(table, 1_000_000, seed=S)
    pad = np.zeros((bc_np.shape[0], stride), dtype=np.uint8); pad[:, :L] = bc_np
    bc = torch.from_numpy(pad).to(dev).repeat(10, 1).contiguous()
    assign = torch.empty((n,), dtype=torch.int32, device=dev)
    for name in libs * 2:
        ctx = seqkit_amd.Context(0, lib_path=os.path.abspath(f"tools/ab/{name}.so"))
        ctx.set_barcodes(table, 1)
        for _ in range(3): ctx.demux_assign_dev(bc.data_ptr(), stride, n, assign.data_ptr())
        ctx.sync(); ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(10): ctx.demux_assign_dev(bc.data_ptr(), stride, n, assign.data_ptr())
            ts.append(ctx.timer_stop() / 10)
        ms = sorted(ts)[2]
        print(f"S={S} L={L} stride={stride}", name, f"{ms * 1e3:.2f} us  {n * (stride + 4) / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
        ctx.close()
