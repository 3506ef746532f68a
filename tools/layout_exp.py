#!/usr/bin/env python3
"""SoA matrices vs the tile-blocked layout for the fused pass, IN ONE PROCESS on the same GPU unit and the same
clusters (units of the pool differ by up to 19 %, so layouts must be compared on one unit).  Checks that both layouts
give the same bytes, then times them alternately.
usage: python tools/layout_exp.py [clusters] [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import seqkit_amd  # noqa: E402
from seqkit_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L, LB = 150, 17
dev = torch.device("cuda", 0)
ctx = seqkit_amd.Context(0)
table = synth.make_sheet(96, 8, dual=True, seed=4)
ctx.set_barcodes(table, 1)
nt = (n + 63) // 64
npad = nt * 64
seq, qual, bc = bench.gen_shard(torch, dev, npad, table, seed=4000, chunk=2_000_000)
out = [torch.empty_like(seq[0]) for _ in range(2)]
lk = [torch.empty((npad,), dtype=torch.int16, device=dev) for _ in range(2)]
assign = torch.empty((npad,), dtype=torch.int32, device=dev)
counts = torch.zeros((99,), dtype=torch.int64, device=dev)
mates = [{"seq": seq[i].data_ptr(), "qual": qual[i].data_ptr(), "len": 0, "out_seq": out[i].data_ptr(), "lowest_k": lk[i].data_ptr()} for i in range(2)]

lay = capi.blocked_layout(2, L, LB, capi.SK_BLK_MASK | capi.SK_BLK_TRIM)
bin_, bout = bench.pack_blocked(torch, lay, seq, qual, bc, nt)
torch.cuda.synchronize()


def run_soa():
    ctx.fused_pass_dev(n, L, 20, mates, bc=bc.data_ptr(), bc_stride=LB, assign=assign.data_ptr(), counts=counts.data_ptr())


def run_blk():
    ctx.fused_pass_blocked_dev(lay, bin_.data_ptr(), bout.data_ptr(), n, 20, counts=counts.data_ptr())


# same bytes from both layouts
counts.zero_(); torch.cuda.synchronize()
run_soa(); ctx.sync()
c_soa = counts.clone()
counts.zero_(); torch.cuda.synchronize()
run_blk(); ctx.sync()
assert torch.equal(c_soa, counts), "counters differ"
u = bench.unpack_blocked(torch, lay, bout, nt)
for i in range(2):
    assert torch.equal(u["out_seq"][i][:n], out[i][:n]), f"out_seq[{i}] differs"
    assert torch.equal(u["lowest_k"][i][:n], lk[i][:n]), f"lowest_k[{i}] differs"
assert torch.equal(u["assign"][:n], assign[:n]), "assign differs"
print("layouts agree on", n, "clusters", flush=True)
del u


def probe(run):
    for _ in range(2):
        run()
    ctx.sync()
    ts = []
    for _ in range(3):
        ctx.timer_start()
        for _ in range(3):
            run()
        ts.append(ctx.timer_stop() / 3)
    return sorted(ts)[1]


for r in range(rounds):
    for name, run in (("soa", run_soa), ("blocked", run_blk)):
        ms = probe(run)
        print(f"round {r} {name:8s}: {ms:7.3f} ms  {925 * n / ms / 1e6 / 80:.1f}% of 8 TB/s", flush=True)
for env, vals in (("SK_TILE_WGS", ["1", "2", "3", "4"]),):
    for v in vals:
        os.environ[env] = v
        os.environ["SK_TILE_WAVES"] = "4"
        print(f"{env}={v}: soa {probe(run_soa):7.3f} ms  blocked {probe(run_blk):7.3f} ms", flush=True)
    os.environ.pop(env); os.environ.pop("SK_TILE_WAVES")
