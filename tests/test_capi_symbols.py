"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from seqkit_amd import capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "seqkit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sk_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/seqkit_hip.h but not exported"


def test_library_carries_gfx950_code_object(hip_lib):
    blob = open(hip_lib, "rb").read()
    assert b"gfx950" in blob
    assert b"tile_pass_kernel" in blob


def test_version_and_no_silent_fallback(hip_lib):
    lib = capi.load_library()
    assert lib.sk_version() >= 0x000100
    # no compute without a GPU: on a box without one sk_create must fail loudly, not fall back
    if lib.sk_device_count() == 0:
        with pytest.raises(capi.SeqkitHipError):
            capi.Context(0)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under seqkit_amd/ may import, link or exec it."""
    bad = []
    for root, _dirs, files in os.walk(os.path.join(REPO, "seqkit_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                if re.search(r"\boracle\b|liboracle|orc_[a-z]", txt):
                    bad.append(os.path.join(root, f))
    assert not bad, bad
