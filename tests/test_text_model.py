"""The text layer of `fasta demultiplex` — both command lines of this repo (the C oracle on the CPU, the HIP host on the GPU box)
against the plain-Python statement of the reference's lines in tests/ref_text_model.py, on inputs drawn by hypothesis around the
places where a shared misreading would hide: headers with several ` BC:` candidates (a first one whose class
byte is invalid, lower case, `+`), non-ASCII text, `trim_end` over U+0085 / U+00A0 / U+2028 / 0x1C-0x1F, `{:.1}` at ties and 0 / 0,
UMI extraction when a barcode holds multi-byte characters (`chars().zip()`), index files, paired mates, dry runs, invalid UTF-8."""
import gzip
import os
import shutil
import subprocess
import tempfile

import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import ref_text_model as model

WS_TAILS = ["", "", " ", "\t", " \t ", " ", "", " ", "　 ", "\x1f", "\x1c ", " \x1e", "\r", "​"]      # (0x1C-0x1F and U+200B are NOT White_Space)
CHARS = ["A", "C", "G", "T", "N", "a", "c", "g", "t", "n", "+", "U", "é", "X", " "]


@st.composite
def cases(draw):
    L = draw(st.integers(2, 7))                                   # barcode length in BYTES

    def bytes_long(chars, want):                                  # characters -> a string of exactly `want` bytes
        s = ""
        for c in chars:
            if len((s + c).encode()) <= want:
                s += c
        return s + "A" * (want - len(s.encode()))

    sheet_alpha = st.sampled_from(["A", "C", "G", "T", "N", "U", "U", "a", "+", "é"])
    n_samples = draw(st.integers(1, 4))
    bcs = [bytes_long(draw(st.lists(sheet_alpha, min_size=1, max_size=L)), L) for _ in range(n_samples)]
    if draw(st.booleans()) and n_samples > 1:
        bcs[1] = bcs[0]                                           # a duplicated barcode: every hit is ambiguous
    names = [f"S{i}" for i in range(n_samples)]
    if draw(st.integers(0, 9)) == 0:
        names[-1] = "Sé"
    lines = []
    for nm, bc in zip(names, bcs):
        lead = draw(st.sampled_from(["", "", "", " ", " "]))
        extra = draw(st.sampled_from(["", "", "\tnote", "\t"]))
        lines.append(lead + nm + "\t" + bc + extra + draw(st.sampled_from(WS_TAILS)) + "\n")
    if draw(st.integers(0, 5)) == 0:
        lines.insert(draw(st.integers(0, len(lines))), draw(st.sampled_from(["# comment\tx\n", "lonely\n", "\n", " \t \n"])))
    sheet = "".join(lines).encode()
    n_reads = draw(st.sampled_from([0, 1, 2, 3, 4, 8, 8, 16, 16]))      # (8 and 16: percentages like 12.5, 6.25, 18.75 — ties of `{:.1}`)
    mode = draw(st.sampled_from(["header", "header", "index1", "index2"]))
    paired = draw(st.booleans())
    obs_alpha = st.sampled_from(CHARS)

    def observed():
        kind = draw(st.integers(0, 9))
        if kind <= 5:                                             # a sheet barcode, maybe with a substitution
            b = list(draw(st.sampled_from(bcs)))
            if kind >= 3 and b:
                b[draw(st.integers(0, len(b) - 1))] = draw(obs_alpha)
            return "".join(b)
        return "".join(draw(st.lists(obs_alpha, min_size=0, max_size=L + 1)))

    def regex_safe(s):                                            # what the header's regex can carry: the class's letters only
        return "".join(c for c in s if c in "ACGTNacgtn+")

    fq1, fq2, idx = [], [], [[], []]
    for i in range(n_reads):
        obs = observed()
        head = f"@r{i}" + draw(st.sampled_from(["", " 1:N:0", " é", " BC:", " BC:X" + "A" * L, " BC", " x"]))
        if mode == "header":
            if draw(st.integers(0, 11)) != 0:
                head += " BC:" + (regex_safe(obs) or "A")
            head += draw(st.sampled_from(["", " extra", " BC:ACGT", " 2:N:0 é"]))
        fq1.append(head + draw(st.sampled_from(WS_TAILS)) + "\n" + "ACGT\n+\nIIII\n")
        if paired:
            h2 = f"@r{i}" + draw(st.sampled_from(["", " 2:N:0", " BC:" + (regex_safe(obs) or "C"), " BC:AC tail", " é BC:GG"])) + draw(st.sampled_from(WS_TAILS))
            fq2.append(h2 + "\n" + "TTTT\n+\n####\n")
        if mode != "header":
            for k in range(1 if mode == "index1" else 2):
                part = obs if mode == "index1" else (obs[:len(obs) // 2] if k == 0 else obs[len(obs) // 2 + 1:])
                idx[k].append(f"@i{i}\n" + part + draw(st.sampled_from(WS_TAILS)) + "\n+\n" + "I" * max(len(part), 1) + "\n")
    files = {"sheet.tsv": sheet, "r1.fq": "".join(fq1).encode()}
    args = []
    if mode != "header":
        files["i1.fq"] = "".join(idx[0]).encode()
        args.append("--index1=i1.fq")
        if mode == "index2":
            files["i2.fq"] = "".join(idx[1]).encode()
            args.append("--index2=i2.fq")
    dry = draw(st.sampled_from([0, 0, 0, 1, 5, 100]))
    if dry:
        args.append(f"--dry-run={dry}")
    args += ["sheet.tsv", "r1.fq"]
    if paired:
        files["r2.fq"] = "".join(fq2).encode()
        args.append("r2.fq")
    if draw(st.integers(0, 14)) == 0 and n_reads:                 # invalid UTF-8 somewhere in the first mate's file
        b = bytearray(files["r1.fq"])
        b[draw(st.integers(0, len(b) - 1))] = 0xFF
        files["r1.fq"] = bytes(b)
    return files, args, mode, paired, dry


def run_case(binary, case, panic_below_100=True):
    files, args, mode, paired, dry = case
    d = tempfile.mkdtemp(prefix="sk_text_")
    try:
        for name, data in files.items():
            with open(os.path.join(d, name), "wb") as f:
                f.write(data)
        r = subprocess.run([binary, "demultiplex"] + args, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        got_files = {f: gzip.open(os.path.join(d, f), "rb").read().decode("utf-8", "replace") for f in sorted(os.listdir(d)) if f.endswith(".gz")}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    index = [files[k] for k in ("i1.fq", "i2.fq") if k in files]
    code, out, err, want_files = model.demultiplex(files["sheet.tsv"], files["r1.fq"], files.get("r2.fq"), index, dry, panic_below_100)
    assert r.returncode == code, (r.returncode, code, r.stderr[-400:], err[-400:])
    if code == 101:
        assert r.stderr.decode("utf-8", "replace").startswith(err)      # (then the panic's own message, which is the runtime's)
        return
    assert r.stderr.decode("utf-8", "replace") == err
    if code == 0:
        assert got_files == want_files
    if dry and code == 0:
        # `- name: count` lines in descending count; equal counts in any order (a HashMap's); this build prints every entry where the
        # reference panics below a hundred
        entries = out[0]
        got = [ln[2:].rsplit(": ", 1) for ln in r.stdout.decode("utf-8", "replace").splitlines()]
        assert sorted((int(c), n) for n, c in got) == sorted((c, n) for n, c in entries)[-len(got):] if len(entries) > 100 else sorted((int(c), n) for n, c in got) == sorted((c, n) for n, c in entries)
        assert [int(c) for _, c in got] == sorted((int(c) for _, c in got), reverse=True)
    else:
        assert r.stdout == b""


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def test_oracle_cli_against_the_python_model(oracle, case):
    run_case(oracle.FASTA_BIN, case)


@pytest.mark.gpu
@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def test_hip_cli_against_the_python_model(hip_lib, case):
    from seqkit_amd import build
    from tests import cli_util as cu
    build.build_hosts()
    run_case(cu.FASTA, case, panic_below_100=False)


@pytest.mark.parametrize("s,want", [("ACGT   \n", "ACGT"), ("ACGT\x1f\n", "ACGT\x1f"), ("ACGT\x1c \x1d", "ACGT\x1c \x1d"), ("é　​", "é　​"), ("", "")])
def test_oracle_trim_end_is_the_unicode_white_space_property(oracle, s, want):
    assert model.trim_end(s) == want
    assert oracle.trim_end_len(s.encode()) == len(want.encode())


@pytest.mark.parametrize("a,b,want", [(1, 8, "12.5"), (1, 16, "6.2"), (3, 16, "18.8"), (5, 16, "31.2"), (7, 16, "43.8"), (1, 3, "33.3"), (2, 3, "66.7"), (0, 0, "NaN"), (5, 5, "100.0"),
                                      (1, 2000, "0.1"), (1, 2001, "0.0"), (199, 2000, "10.0"), (1999, 2000, "100.0")])
def test_model_percentages(a, b, want):
    assert model.fmt1(a / b * 100.0 if b else float("nan")) == want
