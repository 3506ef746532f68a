"""Worker of tests/test_distributed.py: one rank of the sharded demultiplex count reduce on CPU (gloo).
The per-shard arithmetic is done by the CPU oracle here (test infrastructure standing in for the GPU)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import oracle as orc  # noqa: E402
from seqkit_amd import shard, synth  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 10007
    table = synth.make_sheet(96, 8, dual=True, seed=4)
    bc, _ = synth.observe_barcodes(table, n, seed=4, halves=2)          # same seed on every rank = the same pooled run
    flag, tid, mtid, tlen = synth.make_bam_cores(n, seed=5)
    lo, hi = shard.shard_bounds(n, rank, world)
    # every rank learns all bounds and checks the partition
    b = torch.tensor([lo, hi], dtype=torch.int64)
    allb = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allb, b)
    edges = [int(x[0]) for x in allb] + [int(allb[-1][1])]
    assert edges[0] == 0 and edges[-1] == n and all(edges[i] <= edges[i + 1] for i in range(world)), edges
    assert all(int(allb[i][1]) == int(allb[i + 1][0]) for i in range(world - 1))
    assert all(e % shard.TILE_ROWS == 0 for e in edges[:-1])
    # shard-local counters, then the one collective of the path
    local = orc.demux_batch(table, bc[lo:hi], 1)[4]
    counts = torch.from_numpy(local.astype(np.int64))
    shard.reduce_counts(counts)
    whole = orc.demux_batch(table, bc, 1)[4].astype(np.int64)
    assert np.array_equal(counts.numpy(), whole), (counts.numpy(), whole)
    c, h, t = orc.bam_flag_tlen(flag[lo:hi], tid[lo:hi], mtid[lo:hi], tlen[lo:hi], 5000)
    vec = torch.from_numpy(np.concatenate([c, [t], h]).astype(np.int64))
    shard.reduce_counts(vec)
    ec, eh, et = orc.bam_flag_tlen(flag, tid, mtid, tlen, 5000)
    assert np.array_equal(vec.numpy(), np.concatenate([ec, [et], eh]).astype(np.int64))
    dist.barrier()
    if rank == 0:
        print("DIST_OK", world, edges)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
