import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as orc
    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(REPO, "tests", "golden", "appendix_a.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip_lib():
    """libseqkit_hip.so, cross-compiled for gfx950 if it is not there yet (no GPU needed)."""
    from seqkit_amd import build
    build.build_library()
    return build.LIB_PATH


@pytest.fixture(scope="session")
def ctx(hip_lib):
    """A live sk_ctx on GPU 0.  Fails loudly (no fallback) when the GPU or the library is missing."""
    import seqkit_amd
    c = seqkit_amd.Context(0)
    yield c
    c.close()
