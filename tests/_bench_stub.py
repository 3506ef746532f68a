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
if os.environ.get("SK_STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("SK_STUB_FAIL_RANK") is not None:
    time.sleep(30)          # a surviving rank would sit in the rendezvous: the launcher must end it
if rank == 0:
    print("noise on stdout before the line")
    print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "stub": True}))
