"""Read sharding across GPUs (used by bench.py for its shards and its gloo fallback sum): every cluster is independent, so a batch is cut into contiguous,
tile-aligned shards, one per rank, and the only cross-shard state — the additive counters
(src/fasta_demultiplex.rs:108-109,169,177-178; BAM: 3 counters + histogram) — is summed with ONE all-reduce.
One process per GPU; `torch.distributed` backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

TILE_ROWS = 64      # shards start on tile boundaries so that every rank's matrices stay 16-byte aligned


def shard_bounds(n: int, rank: int, world: int, align: int = TILE_ROWS) -> tuple[int, int]:
    """[lo, hi) of `rank`'s contiguous shard of n rows; shards are `align`-row aligned, cover [0, n) exactly once,
    keep input order (rank r rows precede rank r+1 rows) and differ in size by less than two tiles."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    tiles = (n + align - 1) // align
    base, extra = divmod(tiles, world)
    lo_t = rank * base + min(rank, extra)
    hi_t = lo_t + base + (1 if rank < extra else 0)
    return min(lo_t * align, n), min(hi_t * align, n)


def reduce_counts(counts, group=None):
    """Sum the u64[S+3] (or BAM u64[4+bins]) counter vector over all ranks, in place.  `counts` is a torch int64
    tensor on the rank's device; with one rank this is a no-op.  < 40 KB: latency-bound, one call per batch."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts
