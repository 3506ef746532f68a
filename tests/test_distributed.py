"""CPU: the N>1 path — tile-aligned read sharding + the single count all-reduce — with gloo, world_size 2 and 3."""
import os
import subprocess
import sys

import pytest

from seqkit_amd import shard

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(0, 1), (1, 2), (63, 2), (64, 2), (65, 8), (10007, 3), (500_000_000, 8)])
def test_shard_bounds_partition(n, world):
    edges = [shard.shard_bounds(n, r, world) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    for r in range(world - 1):
        assert edges[r][1] == edges[r + 1][0]
        assert edges[r][1] % shard.TILE_ROWS == 0 or edges[r][1] == n
    sizes = [hi - lo for lo, hi in edges]
    assert max(sizes) - min(sizes) < 2 * shard.TILE_ROWS      # one tile of imbalance + the ragged last tile


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_count_reduce_gloo(oracle, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29511 + world), os.path.join(REPO, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:]
