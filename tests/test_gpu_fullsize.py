"""GPU: BASELINE.json's configurations at their full sizes (tools/fullsize_check.py, run in its own process because torch
generates the 58 GB shard on the device and must initialise the HIP runtime before the library does)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg5", "census", "cfg4"])
def test_baseline_config_at_full_size(hip_lib, oracle, cfg):
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fullsize_check.py"), cfg], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=900, cwd=REPO)
    assert r.returncode == 0 and f"{cfg} ok".encode() in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
