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


def test_bam_records_by_block_equal_the_record_by_record_walk(hip_lib, tmp_path):
    """host::BgzfStream::bam_records (ASan + UBSan build): the cores of the records that lie wholly inside a block, walked by
    the thread that inflated it when the block begins with a record, with read()/skip() for the records it leaves — against the
    same stream read record by record, and against the spec decoder: blocks that end on record boundaries (what htslib
    writes), blocks cut anywhere (records and even the 36-byte heads straddle), a record larger than a block, an invalid
    block_size in the middle, a file cut inside a record, a corrupt block; one worker thread and several."""
    from tests.test_cli_cpu import _build_cpp
    exe = _build_cpp(tmp_path, "bgzf_stream_test")
    rng = np.random.default_rng(21)
    recs = []
    for i in range(6000):
        l_seq = int(rng.choice([0, 1, 36, 151, 400]))
        recs.append({"tid": int(rng.integers(-1, 3)), "pos": int(rng.integers(0, 1 << 28)), "flag": int(rng.integers(0, 4096)), "mtid": int(rng.integers(-1, 3)),
                     "mpos": int(rng.integers(0, 1 << 28)), "tlen": int(rng.integers(-6000, 6000)), "name": "r%d" % i * int(rng.integers(1, 4)),
                     "mapq": int(rng.integers(0, 61)), "seq_len": l_seq})
    recs[3000]["seq_len"] = 70_000                                     # larger than a BGZF block
    hdr = b"BAM\1" + struct.pack("<i", 4) + b"@HD\n" + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1 << 28)

    def rec_bytes(r):
        name = r["name"].encode() + b"\0"
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r["mapq"], 4680, 1, r["flag"], r["seq_len"], r["mtid"], r["mpos"], r["tlen"])
        body += name + struct.pack("<I", (r["seq_len"] << 4)) + bytes((r["seq_len"] + 1) // 2) + bytes([30] * r["seq_len"])
        return struct.pack("<i", len(body)) + body
    rb = [rec_bytes(r) for r in recs]
    expect = [f'{r["tid"]} {r["pos"]} {r["flag"]} {r["mtid"]} {r["mpos"]} {r["tlen"]} {r["mapq"]} {len(r["name"]) + 1}' for r in recs]

    def aligned_blocks(chunks):                                            # htslib's bgzf_flush_try: a block is flushed rather than a record split
        out, cur = [], b""
        for c in chunks:
            if cur and len(cur) + len(c) > 0xff00:
                out.append(cur); cur = b""
            cur += c
            while len(cur) > 0xff00:                                       # a record larger than a block does get split
                out.append(cur[:0xff00]); cur = cur[0xff00:]
        if cur:
            out.append(cur)
        return out

    def run(path, *mode, env=None):
        r = subprocess.run([str(exe), str(path), "records", str(len(hdr)), *mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, **(env or {})))
        return r.returncode, r.stdout.decode().splitlines(), r.stderr.decode()

    f = tmp_path / "r.bam"
    # 1. blocks on record boundaries: (nearly) every record comes through bam_records
    f.write_bytes(cu.bgzf_block(hdr) + b"".join(cu.bgzf_block(b) for b in aligned_blocks(rb)) + cu.bgzf_block(b""))
    for env in ({}, {"SEQKIT_THREADS": "1"}, {"SEQKIT_NO_LIBDEFLATE": "1"}, {"SEQKIT_NO_MMAP": "1"}):
        rc, lines, err = run(f, env=env)
        assert rc == 0 and lines == expect + ["end: clean after 6000 records"], err
        assert int(err.split(",")[1].split()[0]) >= 5990, err              # all but the giant record and its neighbours
    assert run(f, "slow")[1] == lines
    # 2. blocks cut anywhere, also inside the header block: the same records
    raw = hdr + b"".join(rb)
    for cut in (60000, 4093, 65280):
        f.write_bytes(b"".join(cu.bgzf_block(raw[i:i + cut]) for i in range(0, len(raw), cut)) + cu.bgzf_block(b""))
        rc, lines, err = run(f)
        assert rc == 0 and lines == expect + ["end: clean after 6000 records"], (cut, err)
    # 3. an invalid block_size in the middle of a block: the records before it, then the error — both ways alike
    bad = list(rb)
    bad[2500] = struct.pack("<i", 31) + bad[2500][4:]
    f.write_bytes(cu.bgzf_block(hdr) + b"".join(cu.bgzf_block(b) for b in aligned_blocks(bad)) + cu.bgzf_block(b""))
    rc, lines, _ = run(f)
    assert rc == 5 and lines == expect[:2500] + ["end: invalid record after 2500 records"]
    assert run(f, "slow")[:2] == (rc, lines)
    # 4. the file cut inside a record
    recs[1000]["seq_len"] = 151
    whole = cu.bgzf_block(hdr) + b"".join(cu.bgzf_block(b) for b in aligned_blocks(rb[:1000] + [rec_bytes(recs[1000])[:50]]))
    f.write_bytes(whole)
    rc, lines, _ = run(f)
    assert rc == 4 and lines == expect[:1000] + ["end: premature after 1000 records"]
    assert run(f, "slow")[:2] == (rc, lines)
    # 5. a corrupt block: what precedes it is delivered, then the stream says so
    blocks = [cu.bgzf_block(hdr)] + [cu.bgzf_block(b) for b in aligned_blocks(rb)]
    k = 5
    broken = bytearray(blocks[k]); broken[30] ^= 0x5a
    f.write_bytes(b"".join(blocks[:k]) + bytes(broken) + b"".join(blocks[k + 1:]))
    rc, lines, _ = run(f)
    rc2, lines2, _ = run(f, "slow")
    assert rc == 3 and (rc2, lines2) == (rc, lines) and lines[-1].startswith("end: invalid after") and lines[:-1] == expect[:len(lines) - 1]
