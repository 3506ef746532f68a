"""bench.py --gpus N without a launcher starts N rank processes itself (before torch / the GPU is touched), hands each
its RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, relays rank 0's JSON line and fails when any rank fails.  CPU only: the
ranks are a stub (--test-worker; the environment cannot redirect the launcher)."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tmp_path, n, extra_env=None, args=()):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update({"SK_STUB_DIR": str(tmp_path)})
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), *args, "--test-worker", os.path.join(REPO, "tests", "_bench_stub.py")],
                          env=env, capture_output=True, text=True, timeout=120)


def test_launcher_starts_n_ranks_with_the_right_environment(tmp_path):
    r = run(tmp_path, 4, args=("--steps", "2", "--warmup", "1"))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"n_gpus": 4, "stub": True}      # exactly one line: rank 0's JSON
    recs = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(4)]
    assert [x["RANK"] for x in recs] == ["0", "1", "2", "3"] and [x["LOCAL_RANK"] for x in recs] == ["0", "1", "2", "3"]
    assert all(x["WORLD_SIZE"] == "4" and x["MASTER_ADDR"] == "127.0.0.1" for x in recs)
    assert len({x["MASTER_PORT"] for x in recs}) == 1 and recs[0]["MASTER_PORT"].isdigit()
    assert all(x["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for x in recs)
    assert all(x["argv"] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] for x in recs)


def test_launcher_fails_when_a_rank_fails_and_ends_the_others(tmp_path):
    t0 = time.time()
    r = run(tmp_path, 3, {"SK_STUB_FAIL_RANK": "1"})
    assert r.returncode == 7 and "rank 1 failed" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 25          # the sleeping ranks were ended, not waited for


def test_a_failing_rank_is_quoted(tmp_path):
    """Every rank's stderr is relayed line by line behind its rank, and the failing rank's last lines are repeated under the verdict."""
    r = run(tmp_path, 3, {"SK_STUB_FAIL_RANK": "2"})
    assert r.returncode == 7
    assert "[rank 2] hipErrorOutOfMemory while allocating the candidates" in r.stderr
    assert "[bench] rank 2 failed with exit code 7" in r.stderr and "[bench]   rank 2 said: the last thing this rank said" in r.stderr


def test_a_taken_port_is_redrawn_even_when_another_rank_fails_first(tmp_path):
    """Only rank 0 binds the rendezvous port.  When it is taken the other ranks can fail first, with an ordinary error: the
    launcher waits a moment for rank 0's own verdict, draws a new port and starts all ranks again."""
    r = run(tmp_path, 3, {"SK_STUB_PORT_TAKEN_ONCE": "1"})
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"n_gpus": 3, "stub": True}
    ports = [open(tmp_path / f"attempt_rank{k}").read().split() for k in range(3)]
    assert all(len(p) == 2 for p in ports) and len({p[1] for p in ports}) == 1       # two attempts, the second on one port for all
    assert "was taken before rank 0 could listen" in r.stderr


def test_under_a_launcher_bench_is_one_rank(tmp_path):
    """With WORLD_SIZE in the environment (torchrun) bench.py does not spawn: it is a rank, and a --gpus that disagrees is an error."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_the_environment_cannot_redirect_the_launcher(tmp_path):
    """SK_BENCH_WORKER (the old test hook) is not honoured: an inherited variable must not make bench.py relay another script's output."""
    env = dict(os.environ, SK_BENCH_WORKER=os.path.join(REPO, "tests", "_bench_stub.py"), SK_STUB_DIR=str(tmp_path), SK_BENCH_LAUNCH_TIMEOUT="60")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert '"stub": true' not in r.stdout and not (tmp_path / "rank0.json").exists()
    assert r.returncode != 0          # the real ranks need a GPU


def rank_run(tmp_path, n, extra_env=None, expect_rc=0):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--test-worker", os.path.join(REPO, "tests", "_bench_rank_worker.py")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == expect_rc, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]) if expect_rc == 0 else r.stderr


def test_ranks_meet_and_agree_on_the_count_reduce(tmp_path):
    """World size 2 / 3 on CPU (gloo), started by bench.py's launcher: every rank says whether it could join, receives rank 0's
    communicator id and joins; when one rank is not ready, or the id cannot be made, ALL ranks fall back to the gloo sum together
    — before anyone enters the (blocking) bootstrap.  The JSON line names the backend, the cost of the reduce and the spread of
    the ranks' kernel times."""
    ok = rank_run(tmp_path, 2)
    assert ok["n_gpus"] == 2 and [x[0] for x in ok["ranks"]] == [None, None]
    assert [x[1] for x in ok["ranks"]] == [True, True] and [x[2] for x in ok["ranks"]] == [0, 1]
    assert [x[3] for x in ok["ranks"]] == [5, 10]                      # RCCL would have summed; the stand-in does not
    assert ok["count_reduce_backend"] == "rccl" and ok["allreduce_us"] == 60.0 and ok["allreduce_us_max"] == 80.0
    assert ok["kernel_ms_ranks"] == {"min": 9.0, "max": 10.0}
    no_uid = rank_run(tmp_path, 2, {"SK_STUB_UID_FAILS": "1"})
    assert all(x[0] == "no interface" for x in no_uid["ranks"]) and [x[3] for x in no_uid["ranks"]] == [15, 15]
    assert no_uid["count_reduce_backend"] == "gloo"
    not_ready = rank_run(tmp_path, 3, {"SK_STUB_NOT_READY_ON": "2"})
    assert all(x[0] == "rank 2: librccl.so.1 cannot be loaded" for x in not_ready["ranks"]) and [x[3] for x in not_ready["ranks"]] == [30, 30, 30]
    assert all(x[2] is None for x in not_ready["ranks"])               # nobody entered the bootstrap
    assert not_ready["count_reduce_backend"] == "gloo" and not_ready["allreduce_us_max"] == 120.0


def test_require_rccl_turns_the_fallback_into_a_failure(tmp_path):
    err = rank_run(tmp_path, 2, {"SK_STUB_UID_FAILS": "1", "SK_BENCH_REQUIRE_RCCL": "1"}, expect_rc=1)
    assert "SK_BENCH_REQUIRE_RCCL=1" in err and "no interface" in err


def test_a_join_that_raises_sends_all_ranks_to_gloo_and_one_that_hangs_ends_the_run(tmp_path):
    """The join is a blocking collective.  A call that RAISES (RCCL refusing the device set) is told to all ranks afterwards and
    all fall back together.  A call that never returns leaves no way back in-process: that rank exits non-zero after
    SK_BENCH_RCCL_TIMEOUT, and bench.py's launcher ends the other ranks instead of waiting for its own timeout."""
    one_out = rank_run(tmp_path, 3, {"SK_STUB_INIT_FAILS_ON": "2"})
    assert all(x[0] == "rank 2 cannot join" for x in one_out["ranks"]) and [x[3] for x in one_out["ranks"]] == [30, 30, 30]
    assert one_out["count_reduce_backend"] == "gloo"
    t0 = time.time()
    err = rank_run(tmp_path, 3, {"SK_STUB_INIT_HANGS_ON": "1", "SK_BENCH_RCCL_TIMEOUT": "3"}, expect_rc=3)
    assert "rank 1: joining the RCCL communicator did not return" in err and "rank 1 failed with exit code 3" in err
    assert time.time() - t0 < 60
