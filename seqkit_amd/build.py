"""Build helpers: compile the gfx950 C-ABI library and the C++ hosts in-tree with hipcc.

The built files live under seqkit_amd/lib/ and seqkit_amd/bin/ (git-ignored, but they
travel to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(ROOT)
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
BINDIR = os.path.join(ROOT, "bin")
LIB_PATH = os.path.join(LIBDIR, "libseqkit_hip.so")

HIP_SOURCES = ["sk_kernels.hip", "sk_census.hip", "sk_inflate.hip", "sk_deflate.hip", "sk_capi.hip", "sk_bamfile.cpp", "sk_lut.cpp"]
HIP_DEPS = HIP_SOURCES + ["sk_internal.h", "sk_lut.h", os.path.join(REPO, "include", "seqkit_hip.h")]
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def library_inputs_digest() -> str:
    """Digest of everything the library is built from: every source and header of HIP_DEPS and the compiler flags.  (The
    built file itself is not reproducible bit for bit — hipcc's object and symbol order vary from run to run — so records
    that must name "this build" name its inputs.)"""
    import hashlib
    h = hashlib.sha256()
    h.update(" ".join(HIP_FLAGS).encode())
    for d in HIP_DEPS:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def _run(cmd: list[str], cwd: str) -> None:
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> seqkit_amd/lib/libseqkit_hip.so (cross-compiles without a GPU)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(LIB_PATH, HIP_DEPS):
        cmd = [_hipcc()] + HIP_FLAGS + ["-o", LIB_PATH] + HIP_SOURCES + ["-ldl", "-lz", "-pthread"]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        _run(cmd, CSRC)
    return LIB_PATH


def build_hosts(force: bool = False) -> list[str]:
    """The C++ `fasta` / `sam` hosts above the C-ABI (seqkit_amd/bin/)."""
    os.makedirs(BINDIR, exist_ok=True)
    out = []
    for name, srcs, libs in (("fasta", ["host_common.cpp", "host_inflate.cpp", "fasta_main.cpp", "fasta_text.cpp"], ["-lz", "-ldl"]),
                             ("sam", ["host_common.cpp", "host_inflate.cpp", "sam_main.cpp"], ["-lz", "-ldl"])):
        if not all(os.path.exists(os.path.join(CSRC, s)) for s in srcs):
            continue
        target = os.path.join(BINDIR, name)
        deps = srcs + ["host_common.h", os.path.join(REPO, "include", "seqkit_hip.h"), LIB_PATH]
        if force or _stale(target, deps):
            cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-pthread", "-I", os.path.join(REPO, "include"),
                   "-o", target] + srcs + ["-L", LIBDIR, "-lseqkit_hip", "-Wl,-rpath,$ORIGIN/../lib"] + libs
            _run(cmd, CSRC)
        out.append(target)
    return out


def build_all(force: bool = False) -> None:
    build_library(force=force)
    build_hosts(force=force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB_PATH)
