"""Stub rank for tests/test_bench_launcher.py: records the environment bench.py's launcher gave it.  No torch, no GPU."""
import json
import os
import sys
import time

rank = int(os.environ["RANK"])
rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
rec["argv"] = sys.argv[1:]
with open(os.path.join(os.environ["SK_STUB_DIR"], f"rank{rank}.json"), "w") as f:
    json.dump(rec, f)
if os.environ.get("SK_STUB_PORT_TAKEN_ONCE"):
    # the first attempt finds its rendezvous port taken: only rank 0 binds it and says so (exit 98, a little later); the other
    # ranks fail first with an ordinary error.  The launcher must still draw a new port; the second attempt runs through.
    marker = os.path.join(os.environ["SK_STUB_DIR"], f"attempt_rank{rank}")
    first = not os.path.exists(marker)
    open(marker, "a").write(os.environ["MASTER_PORT"] + "\n")
    if first:
        if rank == 0:
            time.sleep(1.5)
            sys.exit(98)
        sys.stderr.write("connection refused by 127.0.0.1\n")
        sys.exit(1)
if os.environ.get("SK_STUB_FAIL_RANK") == str(rank):
    sys.stderr.write("hipErrorOutOfMemory while allocating the candidates\nthe last thing this rank said\n")
    sys.exit(7)
if os.environ.get("SK_STUB_FAIL_RANK") is not None:
    time.sleep(30)          # a surviving rank would sit in the rendezvous: the launcher must end it
if rank == 0:
    print("noise on stdout before the line")
    print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "stub": True}))
