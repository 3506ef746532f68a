"""Rank of tests/test_bench_launcher.py: started by bench.py's own launcher (--test-worker), it runs bench.py's rendezvous
code on CPU — gloo process group from the environment the launcher gave, join_count_reduce with stand-ins for the three C-ABI
calls, the gloo fallback sum, and the keys bench.py derives for the JSON line of an N > 1 run.  No GPU."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
joined = {}


def ready():
    if os.environ.get("SK_STUB_NOT_READY_ON") == str(rank):
        raise RuntimeError("librccl.so.1 cannot be loaded")


def make_uid():
    if os.environ.get("SK_STUB_UID_FAILS"):
        raise RuntimeError("no interface")
    return bytes(range(128))


def init_rank(uid, r, w):
    if os.environ.get("SK_STUB_INIT_FAILS_ON") == str(r):
        raise RuntimeError(f"rank {r} cannot join")
    if os.environ.get("SK_STUB_INIT_HANGS_ON") == str(r):
        time.sleep(600)                              # the real call blocks in the bootstrap
    joined["uid"], joined["rank"], joined["world"] = uid, r, w


err = bench.join_count_reduce(dist, rank, world, make_uid, init_rank, ready=ready)
bench.require_backend(err)
counts = torch.arange(99, dtype=torch.int64) * (rank + 1)
if err is not None:                                  # what bench.py's step() does without RCCL
    dist.all_reduce(counts)
total = [None] * world
dist.all_gather_object(total, (err, joined.get("uid") == bytes(range(128)), joined.get("rank"), int(counts[5])))
both = [None] * world
dist.all_gather_object(both, (9.0 + rank, 40.0 * (rank + 1)))       # (kernel_ms, reduce_us) of each rank, as bench.py gathers them
dist.barrier()
if rank == 0:
    line = {"n_gpus": world, "ranks": total}
    line.update(bench.reduce_report(True, err, [b[0] for b in both], [b[1] for b in both]))
    print(json.dumps(line))
dist.destroy_process_group()
