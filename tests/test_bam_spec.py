"""B1 (src/common.rs:121-157, htslib in the reference): the BAM decode of BOTH command-line readers — the product's
(seqkit_amd/csrc/sam_main.cpp) and the oracle's (oracle/sam_oracle_main.c) — against a third decode written from the
SAM/BAM specification alone (tests/bam_spec.py), on the BAM files the CLI tests write and on a BGZF file whose blocks
carry an extra gzip subfield (legal per RFC 1952, never produced by the tests' own writer)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from seqkit_amd import synth
from tests import bam_spec
from tests import cli_util as cu


def make_records(n, seed):
    rng = np.random.default_rng(seed)
    flag, tid, mtid, tlen = synth.make_bam_cores(n, seed=seed)
    recs = []
    for i in range(n):
        wild = rng.random() < 0.2
        recs.append(dict(tid=int(tid[i]) % 3, mtid=(int(mtid[i]) % 3 if mtid[i] >= 0 else -1), pos=int(rng.integers(0, 900000)),
                         mpos=int(rng.integers(0, 900000)), flag=int(rng.integers(0, 4096)) if wild else int(flag[i]),
                         tlen=int(rng.choice([0, 1, -1, 5000, 5001, -5000, -2**31, 2**31 - 1])) if wild else int(tlen[i]),
                         name=f"read{i}", seq_len=int(rng.integers(0, 70)), mapq=int(rng.integers(0, 256)),
                         cigar=[(0, 5)] * int(rng.integers(0, 4))))
    return recs


def decode_matches_writer(recs, got):
    assert len(got) == len(recs)
    for r, g in zip(recs, got):
        assert (g["refID"], g["pos"], g["flag"], g["next_refID"], g["next_pos"], g["tlen"], g["mapq"]) == \
               (r["tid"], r["pos"], r["flag"], r["mtid"], r["mpos"], r["tlen"], r["mapq"])


def expected_outputs(recs):
    return {("statistics",): bam_spec.statistics_text(recs),
            ("fragment", "lengths"): bam_spec.fragment_lengths_text(recs),
            ("fragment", "lengths", "--max-frag-size=300"): bam_spec.fragment_lengths_text(recs, 300),
            ("fragment", "lengths", "--reads=25", "--max-frag-size=700"): bam_spec.fragment_lengths_text(recs, 700, 25)}


def check_binary(binary, bam, recs, tmp_path):
    for args, want in expected_outputs(recs).items():
        rc, out, err = cu.run(binary, list(args) + [str(bam)], cwd=tmp_path)
        assert rc == 0, (args, err)
        assert out == want, args


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_cli_against_the_spec_decode(oracle, tmp_path, seed):
    recs = make_records(3000, seed)
    bam = tmp_path / "a.bam"
    cu.write_bam(str(bam), [("chr1", 1_000_000), ("chr2", 900_000), ("chrM", 16_000)], recs)
    refs, got = bam_spec.read_bam(str(bam))
    assert refs == [("chr1", 1_000_000), ("chr2", 900_000), ("chrM", 16_000)]
    decode_matches_writer(recs, got)
    check_binary(oracle.SAM_BIN, bam, got, tmp_path)


def rewrap_with_extra_subfield(src, dst):
    """The same BGZF payloads, every block re-wrapped with ANOTHER extra subfield ('X','Y', 3 bytes) in front of 'BC'
    (RFC 1952 allows any number of subfields; htslib reads such files)."""
    data = open(src, "rb").read()
    out = b""
    for raw in bam_spec.bgzf_blocks(data):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        extra = struct.pack("<BBH", 88, 89, 3) + b"abc" + struct.pack("<BBH", 66, 67, 2)
        xlen = len(extra) + 2
        bsize = 12 + xlen + len(comp) + 8 - 1
        out += struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen) + extra + struct.pack("<H", bsize) + comp + \
            struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    open(dst, "wb").write(out)


def test_oracle_cli_reads_bgzf_with_other_extra_subfields(oracle, tmp_path):
    recs = make_records(500, 9)
    a, b = tmp_path / "a.bam", tmp_path / "b.bam"
    cu.write_bam(str(a), [("chr1", 1_000_000), ("chr2", 900_000), ("chrM", 16_000)], recs)
    rewrap_with_extra_subfield(a, b)
    _, got = bam_spec.read_bam(str(b))
    decode_matches_writer(recs, got)
    check_binary(oracle.SAM_BIN, b, got, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_product_cli_against_the_spec_decode(hip_lib, tmp_path, seed):
    from seqkit_amd import build
    build.build_hosts()
    recs = make_records(3000, seed)
    bam = tmp_path / "a.bam"
    cu.write_bam(str(bam), [("chr1", 1_000_000), ("chr2", 900_000), ("chrM", 16_000)], recs)
    _, got = bam_spec.read_bam(str(bam))
    check_binary(cu.SAM, bam, got, tmp_path)
    b = tmp_path / "b.bam"
    rewrap_with_extra_subfield(bam, b)
    check_binary(cu.SAM, b, got, tmp_path)


def test_bgzf_stream_walks_blocks_with_other_extra_subfields(hip_lib, tmp_path):
    """host::BgzfStream (ASan + UBSan build) finds 'BC' behind another subfield, in the first block and in later ones, and
    still inflates the blocks in parallel; a header cut inside its extra field ends the data like any other cut."""
    from tests.test_cli_cpu import _build_cpp
    exe = _build_cpp(tmp_path, "bgzf_stream_test")
    rng = np.random.default_rng(8)
    raw = bytes(rng.integers(0, 9, size=900_000, dtype=np.uint8))
    plain = tmp_path / "p.bgzf"
    plain.write_bytes(b"".join(cu.bgzf_block(raw[i:i + 50000]) for i in range(0, len(raw), 50000)) + cu.bgzf_block(b""))
    extra = tmp_path / "x.bgzf"
    rewrap_with_extra_subfield(plain, extra)

    def run(path):
        r = subprocess.run([str(exe), str(path)], stdout=subprocess.PIPE, timeout=120)
        return r.returncode, r.stdout
    assert run(extra) == (0, raw)
    mixed = tmp_path / "m.bgzf"                              # first block plain (parallel path chosen at open), later ones with the subfield
    first = cu.bgzf_block(raw[:50000])
    rest = tmp_path / "rest.bgzf"
    rest.write_bytes(b"".join(cu.bgzf_block(raw[i:i + 50000]) for i in range(50000, len(raw), 50000)))
    rewrap_with_extra_subfield(rest, mixed)
    mixed.write_bytes(first + mixed.read_bytes())
    assert run(mixed) == (0, raw)
    data = extra.read_bytes()
    blocks = list(bam_spec.bgzf_blocks(data))
    cut = tmp_path / "c.bgzf"
    first_len = struct.unpack_from("<H", data, 12 + 7 + 4)[0] + 1          # BSIZE of block 0 (behind the 7-byte XY subfield)
    cut.write_bytes(data[:first_len + 15])                               # block 1 cut inside its extra field
    assert run(cut) == (0, blocks[0])
