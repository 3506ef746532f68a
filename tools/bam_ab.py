#!/usr/bin/env python3
"""BAM flag/TLEN reduction and `sam fragments` filter, this build against other builds of the same C-ABI on the same
columns in one process.  usage: SK_LIBS=tools/ab/x.so python tools/bam_ab.py [records]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import seqkit_amd  # noqa: E402
from seqkit_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
dev = torch.device("cuda", 0)
libs = [("cur", None)] + [(os.path.basename(p)[:-3], os.path.abspath(p)) for p in os.environ.get("SK_LIBS", "").split(",") if p]
ctxs = [(name, seqkit_amd.Context(0, lib_path=path)) for name, path in libs]
flag_np, tid_np, mtid_np, tlen_np = synth.make_bam_cores(2_000_000, seed=5)
reps = n // 2_000_000
flag = torch.from_numpy(flag_np.view("int16")).to(dev).repeat(reps)
tid = torch.from_numpy(tid_np).to(dev).repeat(reps)
mtid = torch.from_numpy(mtid_np).to(dev).repeat(reps)
tlen = torch.from_numpy(tlen_np).to(dev).repeat(reps)
n = flag.numel()
out = torch.zeros(4 + 5001, dtype=torch.int64, device=dev)
bits = torch.empty((n + 7) // 8, dtype=torch.uint8, device=dev)
kept = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for what in ("flag/TLEN reduction", "fragments filter"):
    for name, ctx in ctxs:
        def run():
            if what == "flag/TLEN reduction":
                ctx.bam_flag_tlen_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 5000, out.data_ptr())
            else:
                ctx.bam_fragments_dev(flag.data_ptr(), tid.data_ptr(), mtid.data_ptr(), tlen.data_ptr(), n, 0, 5000, bits.data_ptr(), kept.data_ptr())
        run(); ctx.sync()
        ts = []
        for _ in range(5):
            ctx.timer_start()
            for _ in range(3):
                run()
            ts.append(ctx.timer_stop() / 3)
        ms = sorted(ts)[2]
        b = 14.125 if what == "fragments filter" else 14
        print(f"{what:22s} {name:8s}: {ms:7.4f} ms  {n / ms / 1e6:7.2f} G records/s  {b * n / ms / 1e6 / 80:5.1f}% of 8 TB/s", flush=True)
