"""Rank of tests/test_bench_launcher.py::test_ranks_meet_and_agree_on_the_count_reduce: started by bench.py's own launcher
(SK_BENCH_WORKER), it runs bench.py's rendezvous code on CPU — gloo process group from the environment the launcher gave,
join_count_reduce with stand-ins for the two C-ABI calls — and the gloo fallback sum.  No GPU."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
joined = {}


def make_uid():
    if os.environ.get("SK_STUB_UID_FAILS"):
        raise RuntimeError("no interface")
    return bytes(range(128))


def init_rank(uid, r, w):
    if os.environ.get("SK_STUB_INIT_FAILS_ON") == str(r):
        raise RuntimeError(f"rank {r} cannot join")
    joined["uid"], joined["rank"], joined["world"] = uid, r, w


err = bench.join_count_reduce(dist, rank, world, make_uid, init_rank)
counts = torch.arange(99, dtype=torch.int64) * (rank + 1)
if err is not None:                                  # what bench.py's step() does without RCCL
    dist.all_reduce(counts)
total = [None] * world
dist.all_gather_object(total, (err, joined.get("uid") == bytes(range(128)), joined.get("rank"), int(counts[5])))
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "ranks": total}))
dist.destroy_process_group()
