"""CPU (hipcc cross-compiles): the census front kernel loads its table entries with volatile-asm `ds_read_b128`s so that the
chunk with the state is read FIRST (sk_census.hip, "the workgroup's front table") — loads the compiler does not know of.  What
keeps that sound is that no instruction touches a register such a load writes before an `s_waitcnt lgkmcnt(0)`: checked here in
the ISA of every instantiation, together with "no scratch" in any of them."""
import os
import re
import subprocess

import pytest

from seqkit_amd import build


@pytest.fixture(scope="module")
def census_isa(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "census.s"
    try:
        hipcc = build._hipcc()
    except RuntimeError as e:                           # no compiler on this machine: nothing to check (the build check fails elsewhere)
        pytest.skip(str(e))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        "-o", str(out), os.path.join(build.CSRC, "sk_census.hip")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    return out.read_text(), r.stdout


def test_asm_loads_are_waited_for_before_their_registers_are_touched(census_isa):
    text, _ = census_isa
    names = re.findall(r"^(_ZN2sk13census_kernelILi\dELi\dELb[01]EEEvNS_10CensusArgsEii):", text, re.M)
    assert len(names) == 12
    for name in names:
        i = text.index("\n" + name + ":")
        body = text[i:text.index("s_endpgm", i)].split("\n")
        pending, in_asm, n_asm = {}, False, 0
        for line in (x.strip() for x in body):
            if line.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if line.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not line or line[0] in ";." or line.endswith(":"):
                continue
            if in_asm and line.startswith("ds_read_b128"):
                m = re.match(r"ds_read_b128 v\[(\d+):(\d+)\]", line)
                for reg in range(int(m.group(1)), int(m.group(2)) + 1):
                    pending[reg] = line
                n_asm += 1
                continue
            if line.startswith("s_waitcnt") and "lgkmcnt(0)" in line:
                pending.clear()
                continue
            regs = set()
            for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            regs.update(int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", line))
            assert not (regs & pending.keys()), f"{name}: `{line}` touches a register an asm load writes, before the wait"
        assert n_asm >= 6, (name, n_asm)


def test_no_variant_keeps_anything_in_scratch(census_isa):
    """census_add takes one or two tiles per step (sk_census.hip: kCensusMaxSub; three and four were dropped in round 5 — the same
    time, and scratch for the longest strings): every instantiation has ScratchSize 0 — a scratch reload is a VMEM operation and
    waits for the step's prefetch."""
    _, remarks = census_isa
    scratch = {}
    for m in re.finditer(r"Function Name: _ZN2sk13census_kernelILi(\d)ELi(\d)ELb([01])EEEvNS_10CensusArgsEii.*?ScratchSize \[bytes/lane\]: (\d+)", remarks, re.S):
        scratch[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = int(m.group(4))
    assert len(scratch) == 12
    for key, b in scratch.items():
        assert b == 0, (key, b)
