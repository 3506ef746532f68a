"""CPU: the C++ `fasta` host against the oracle CLI for the parts that need no GPU (argument grammar, exit codes,
`fasta add barcode`, stream plumbing).  Anything arithmetic is GPU-only and lives in test_cli_gpu.py."""
import gzip
import zlib

import numpy as np
import pytest

from tests import cli_util as cu


@pytest.fixture(scope="module")
def bins(hip_lib, oracle):
    from seqkit_amd import build
    build.build_hosts()
    return cu.FASTA, oracle.FASTA_BIN


def same(bins, args, tmp_path, stdin=None):
    a = cu.run(bins[0], args, cwd=tmp_path, stdin=stdin)
    b = cu.run(bins[1], args, cwd=tmp_path, stdin=stdin)
    assert a[0] == b[0], (a, b)
    assert a[1] == b[1]
    return a, b


def test_add_barcode_fastq_fasta_and_stale_barcode(bins, tmp_path):
    fq = tmp_path / "r.fq"
    ix = tmp_path / "i.fq"
    fq.write_bytes(b"@r1 1:N:0\nACGT\n+\nIIII\n@r2 \t\nTTTT\n+x\n####\n@r3\nGG\n+\nII\n")
    ix.write_bytes(b"@i1\nACGTACGT\n+\nIIIIIIII\n@i2\nTTTTGGGG  \n+\nIIIIIIII\n")          # shorter than the reads: last barcode repeats
    a, _ = same(bins, ["add", "barcode", str(fq), str(ix)], tmp_path)
    assert a[1] == b"@r1 1:N:0 BC:ACGTACGT\nACGT\n+\nIIII\n@r2 BC:TTTTGGGG\nTTTT\n+x\n####\n@r3 BC:TTTTGGGG\nGG\n+\nII\n"
    fa = tmp_path / "r.fa"
    ia = tmp_path / "i.fa"
    fa.write_bytes(b">s1\nACGT\n>s2\nTT\n")
    ia.write_bytes(b">b1\nAAAA\n>b2\nCCCC\n")
    same(bins, ["add", "barcode", str(fa), str(ia)], tmp_path)


def bgzf_bytes(data, block=5000):
    """`data` as BGZF (SAMv1 section 4.1), written from the specification: blocks of `block` input bytes and the EOF block."""
    import struct
    out = b""
    for o in list(range(0, len(data), block)) + [None]:
        piece = b"" if o is None else data[o:o + block]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(piece) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(piece), len(piece))
    return out


def test_gz_inputs_plain_gzip_and_bgzf(bins, tmp_path):
    """A *.gz input is inflated whether it is one gzip stream, several members, or BGZF (which the block-parallel reader
    takes): the same output as for the text, in both line readers' commands."""
    rng = np.random.default_rng(11)
    recs = b"".join(b"@r%d some text\n" % i + bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 200))).astype(np.uint8)) + b"\n+\n" + b"I" * 3 + b"\n"
                    for i in range(4000))
    idx = b"".join(b"@r%d\n" % i + bytes(rng.choice(list(b"ACGT"), size=8).astype(np.uint8)) + b"\n+\nIIIIIIII\n" for i in range(4000))
    (tmp_path / "r.fq").write_bytes(recs)
    (tmp_path / "i.fq").write_bytes(idx)
    ref = cu.run(bins[0], ["add", "barcode", "r.fq", "i.fq"], cwd=tmp_path)
    assert ref[0] == 0 and ref[1].count(b" BC:") == 4000
    with gzip.open(tmp_path / "r1.fq.gz", "wb") as f:
        f.write(recs)
    (tmp_path / "r2.fq.gz").write_bytes(gzip.compress(recs[:100000]) + gzip.compress(recs[100000:]))      # two members
    (tmp_path / "r3.fq.gz").write_bytes(bgzf_bytes(recs))
    (tmp_path / "i3.fq.gz").write_bytes(bgzf_bytes(idx, block=777))
    for r, i in (("r1.fq.gz", "i.fq"), ("r2.fq.gz", "i.fq"), ("r3.fq.gz", "i.fq"), ("r3.fq.gz", "i3.fq.gz"), ("r.fq", "i3.fq.gz")):
        got = cu.run(bins[0], ["add", "barcode", r, i], cwd=tmp_path)
        assert got[0] == 0 and got[1] == ref[1], (r, i)
    # a BGZF file cut off inside a block ends there, like a gzip stream that `gunzip -c` could not finish
    whole = bgzf_bytes(recs)
    (tmp_path / "cut.fq.gz").write_bytes(whole[:len(whole) // 2])
    got = cu.run(bins[0], ["add", "barcode", "cut.fq.gz", "i.fq"], cwd=tmp_path)
    assert ref[1].startswith(got[1][:got[1].rfind(b"\n@") + 1]) and 0 < len(got[1]) < len(ref[1])


def test_add_barcode_stdin_gz_and_invalid_line(bins, tmp_path):
    ix = tmp_path / "i.fq.gz"
    with gzip.open(ix, "wb") as f:
        f.write(b"@i1\nACGT\n+\nIIII\n")
    same(bins, ["add", "barcode", "-", str(ix)], tmp_path, stdin=b"@r1\nAC\n+\nII\n")
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r1\nAC\n+\nII\nXoops\nAC\n")
    a, b = same(bins, ["add", "barcode", str(bad), str(ix)], tmp_path)
    assert a[0] == 255 and a[2] == b[2] and b"ERROR: Invalid FASTQ line:" in a[2]


@pytest.mark.parametrize("args", [
    ["trim", "by", "quality"], ["trim", "by", "quality", "x.fq"], ["trim", "by", "quality", "a", "b", "c"],
    ["mask", "by", "quality", "x.fq"], ["add", "barcode", "x"], ["add", "barcode", "x", "y", "z"],
    ["demultiplex"], ["demultiplex", "sheet"], ["demultiplex", "--bogus", "s", "f"], ["demultiplex", "-x", "s", "f"],
    ["demultiplex", "--index1", "s", "f"], ["demultiplex", "a", "b", "c", "d"], ["demultiplex", "--parallel=1", "s", "f"],
])
def test_invalid_arguments_exit_255_with_usage(bins, tmp_path, args):
    a, b = same(bins, args, tmp_path)
    assert a[0] == 255 and a[2] == b[2] and a[2].startswith(b"ERROR: Invalid arguments.\n")


def test_missing_file_and_bad_numbers(bins, tmp_path):
    a, b = same(bins, ["trim", "by", "quality", "nope.fq", "20"], tmp_path)
    assert a[0] == 255 and a[2] == b[2] == b"ERROR: Cannot open file nope.fq for reading.\n"
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r\nA\n+\nI\n")
    for bad in ("256", "2x", "", " 20"):
        ra = cu.run(bins[0], ["trim", "by", "quality", str(fq), bad], cwd=tmp_path)
        rb = cu.run(bins[1], ["trim", "by", "quality", str(fq), bad], cwd=tmp_path)
        assert ra[0] == rb[0] == 101, (bad, ra, rb)                       # parse().unwrap() panics
    sheet = tmp_path / "s.tsv"
    sheet.write_bytes(b"A\tACGT\n")
    for bad in ("0", "x", "-3"):
        a, b = same(bins, ["demultiplex", "--dry-run=" + bad, str(sheet), str(fq)], tmp_path)
        assert a[0] == 255 and a[2] == b[2] == b"ERROR: In --dry-run=N, N must be 64-bit positive integer.\n"


def test_top_level_usage(bins, tmp_path):
    a, b = same(bins, ["frobnicate"], tmp_path)
    assert a[0] == 0 and a[2] == b[2]


def test_sample_sheet_errors(bins, tmp_path):
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r BC:ACGT\nA\n+\nI\n")
    cases = {
        b"A\tACGT\nB\tACG\n": b"ERROR: Barcodes in sample sheet must all be of same length.\n",
        b"A\t\tx\n": b"ERROR: Sample A has no barcode.\n",
    }
    for text, msg in cases.items():
        sheet = tmp_path / "s.tsv"
        sheet.write_bytes(text)
        a, b = same(bins, ["demultiplex", "--dry-run=5", str(sheet), str(fq)], tmp_path)
        assert a[0] == 255 and a[2].endswith(msg) and a[2] == b[2]


def test_block_parallel_gzip_writer(hip_lib, tmp_path):
    """host::GzWriter: many files written interleaved through the compression pool decompress to exactly what was written
    (multi-member gzip, order kept per file), including an empty file."""
    import gzip
    import os
    import subprocess
    from seqkit_amd import build
    exe = tmp_path / "gz_writer_test"
    csrc = build.CSRC
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(build.REPO, "include"), "-I", csrc, "-o", str(exe),
                    os.path.join(build.REPO, "tests", "cpp", "gz_writer_test.cpp"), os.path.join(csrc, "host_common.cpp"), os.path.join(csrc, "host_inflate.cpp"),
                    "-L", build.LIBDIR, "-lseqkit_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lz", "-ldl"], check=True)
    for nf, total in ((5, 3_000_000), (2, 0), (40, 100_000)):
        d = tmp_path / f"o{nf}_{total}"
        d.mkdir()
        env = dict(os.environ)
        if nf == 40:
            env["SEQKIT_NO_LIBDEFLATE"] = "1"           # the zlib fallback
        subprocess.run([str(exe), str(d), str(nf), str(total)], check=True, timeout=120, env=env)
        for i in range(nf):
            got = gzip.open(d / f"f{i}.gz", "rb").read()
            # the file is BGZF (SAMv1 section 4.1): every member carries its own size in a 'BC' extra field, holds at most
            # 64 KiB and inflates on its own; the last one is the empty end-of-file block
            raw = (d / f"f{i}.gz").read_bytes()
            o, parts = 0, []
            while o < len(raw):
                assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and raw[o + 10:o + 16] == b"\x06\x00BC\x02\x00", (i, o)
                bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
                assert 28 <= bsize <= 65536 and o + bsize <= len(raw)
                piece = zlib.decompress(raw[o + 18:o + bsize - 8], -15)
                assert zlib.crc32(piece) == int.from_bytes(raw[o + bsize - 8:o + bsize - 4], "little")
                assert len(piece) == int.from_bytes(raw[o + bsize - 4:o + bsize], "little") <= 65536
                parts.append(piece)
                o += bsize
            assert parts and parts[-1] == b"" and raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
            assert b"".join(parts) == got
            exp = bytes((ord("A") + (k * (i + 3)) % 23) for k in range(total)) if total <= 200_000 else None
            assert len(got) == total
            if exp is not None:
                assert got == exp
            else:
                import numpy as np
                k = np.arange(total, dtype=np.int64)
                assert np.array_equal(np.frombuffer(got, dtype=np.uint8), (65 + (k * (i + 3)) % 23).astype(np.uint8))


def _build_cpp(tmp_path, name, extra=()):
    import os
    import subprocess
    from seqkit_amd import build
    exe = tmp_path / name
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", os.path.join(build.REPO, "include"), "-I", build.CSRC, "-o", str(exe),
                    os.path.join(build.REPO, "tests", "cpp", name + ".cpp"), os.path.join(build.CSRC, "host_common.cpp"), os.path.join(build.CSRC, "host_inflate.cpp"),
                    "-L", build.LIBDIR, "-lseqkit_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lz", "-ldl", *extra], check=True)
    return exe


def test_host_text_helpers_vs_oracle_under_asan(hip_lib, oracle, tmp_path):
    """trim_end / trim / UTF-8 validation / ` BC:` scanner: the host's implementation (ASan+UBSan build) against the oracle's,
    on random byte strings biased towards whitespace, multi-byte characters and BC fields."""
    import random
    import subprocess
    exe = _build_cpp(tmp_path, "host_text_test")
    rnd = random.Random(1234)
    code_points = [0x85, 0xA0, 0x1680, 0x2000, 0x200A, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000, 0x200B, 0xE9, 0x20AC, 0x1F600]
    controls = [9, 10, 13, 11, 12, 0x1C, 0x1F, 32]
    pieces = [bytes([c]) for c in controls] + [chr(c).encode() for c in code_points] + \
             [bytes([0xFF]), bytes([0xC0, 0xAF]), bytes([0xE2, 0x82]), bytes([0xED, 0xA0, 0x80]), bytes([0xF4, 0x90, 0x80, 0x80]),
              b" BC:", b" BC:ACGT", b"BC:", b"ACGTN+acgtn", b"@r1", b"x", b"A", b":", b" "]
    cases = [b"", b" ", bytes([10]), b" BC:", b" BC:A", b"@r BC:X BC:ACGT+TT extra" + bytes([10])]
    for _ in range(4000):
        cases.append(b"".join(rnd.choice(pieces) for _ in range(rnd.randint(0, 9))))
    for _ in range(1000):
        cases.append(bytes(rnd.randrange(256) for _ in range(rnd.randint(0, 12))))
    out = subprocess.run([str(exe)], input=chr(10).join(c.hex() for c in cases).encode() + bytes([10]), stdout=subprocess.PIPE, check=True,
                         env={"ASAN_OPTIONS": "detect_leaks=0"}).stdout.decode().splitlines()
    assert len(out) == len(cases)
    for c, line in zip(cases, out):
        ok, te, ts, asc, b0, b1 = map(int, line.split())
        assert bool(ok) == oracle.utf8_valid(c), c
        assert bool(asc) == all(x < 128 for x in c)
        if ok:
            assert te == oracle.trim_end_len(c), c
            assert ts == int(oracle.lib().orc_trim_start_off(c, len(c))), c
        hit = oracle.find_bc_field(c)
        assert (b0, b1) == (hit if hit else (-1, -1)), c


def test_oracle_cli_under_asan(oracle, tmp_path, golden):
    """The oracle CLIs rebuilt with ASan+UBSan run the golden flows clean (the checker itself must be sound)."""
    import os
    import shutil
    import subprocess
    NL = bytes([10])
    TAB = bytes([9])
    src = os.path.dirname(oracle.LIB)
    d = tmp_path / "orc_asan"
    d.mkdir()
    for f in ("seqkit_oracle.c", "seqkit_oracle.h", "cli_common.h", "fasta_oracle_main.c", "sam_oracle_main.c", "Makefile"):
        shutil.copy(os.path.join(src, f), d / f)
    subprocess.run(["make", "-C", str(d), "CFLAGS=-O1 -g -std=c11 -D_GNU_SOURCE -fPIC -Wno-unused-function -fsanitize=address,undefined -fno-sanitize-recover=undefined",
                    "fasta_oracle", "sam_oracle"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    g = golden["trim_by_quality"]
    fq = tmp_path / "k.fq"
    fq.write_bytes(b"".join(b"@r%d" % i + NL + c["seq"].encode("latin-1") + NL + b"+" + NL + c["qual"].encode("latin-1") + NL for i, c in enumerate(g["cases"])) + b"@eof" + NL + b"AC")
    for cmd in (["trim", "by", "quality", str(fq), "20"], ["mask", "by", "quality", str(fq), "20"], ["add", "barcode", str(fq), str(fq)]):
        a = subprocess.run([str(d / "fasta_oracle")] + cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        b = subprocess.run([oracle.FASTA_BIN] + cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert a.returncode == b.returncode and a.stdout == b.stdout, (cmd, a.stderr[-400:])
    sheet = tmp_path / "s.tsv"
    sheet.write_bytes(b"A" + TAB + b"ACGTUUUU" + NL + b"B" + TAB + b"TTTTUUUU" + NL)
    fq.write_bytes(NL.join([b"@r1 BC:ACGTTTGA", b"ACGT", b"+", b"IIII", b"@r2 BC:GGGGGGGG x", b"AC", b"+", b"II", b""]))
    w = tmp_path / "w"
    w.mkdir()
    a = subprocess.run([str(d / "fasta_oracle"), "demultiplex", str(sheet), str(fq)], cwd=w, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and b"1 / 2 (50.0%)" in a.stderr, a.stderr
    bam = tmp_path / "t.bam"
    cu.write_bam(str(bam), [("chr1", 1000)], [dict(tid=0, mtid=0, flag=99, tlen=180, pos=1, mpos=100)])
    a = subprocess.run([str(d / "sam_oracle"), "fragment", "lengths", "--max-frag-size=200", str(bam)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0 and (b"180" + TAB + b"1" + NL) in a.stdout, a.stderr


def test_bgzf_stream_parallel_inflate_unit(tmp_path):
    """host::BgzfStream: blocks inflated by several threads come out in file order; a file cut inside a block ends the
    data before that block; a flipped byte is reported as corrupt; plain (non-BGZF) gzip still reads; libdeflate and zlib
    paths agree."""
    import gzip
    import os
    import subprocess
    import numpy as np
    exe = _build_cpp(tmp_path, "bgzf_stream_test")
    rng = np.random.default_rng(5)
    raw = bytes(rng.integers(0, 7, size=3_000_000, dtype=np.uint8)) + bytes(rng.integers(0, 256, size=200_000, dtype=np.uint8))
    blocks = [cu.bgzf_block(raw[i:i + 60000]) for i in range(0, len(raw), 60000)] + [cu.bgzf_block(b"")]
    f = tmp_path / "x.bgzf"

    def run(env=None):
        r = subprocess.run([str(exe), str(f)], stdout=subprocess.PIPE, timeout=120, env=dict(os.environ, **(env or {})))
        return r.returncode, r.stdout

    f.write_bytes(b"".join(blocks))
    for env in ({}, {"SEQKIT_NO_LIBDEFLATE": "1"}, {"SEQKIT_THREADS": "1"}, {"SEQKIT_THREADS": "3"}, {"SEQKIT_NO_MMAP": "1"},      # mapped file / read through the descriptor
                {"SEQKIT_NO_LIBDEFLATE": "1", "SEQKIT_ZLIB_INFLATE": "1"}):                                                         # this build's decoder and CRC / zlib's
        assert run(env) == (0, raw)
    whole = b"".join(blocks)
    cut = len(b"".join(blocks[:20])) + 100                                  # inside block 20
    f.write_bytes(whole[:cut])
    assert run() == (0, raw[:20 * 60000])
    f.write_bytes(whole[:len(b"".join(blocks[:20])) + 7])                   # inside a block header
    assert run() == (0, raw[:20 * 60000])
    bad = bytearray(whole)
    bad[len(b"".join(blocks[:10])) + 40] ^= 0x55                            # payload of block 10
    f.write_bytes(bytes(bad))
    rc, out = run()
    assert rc == 3 and out == raw[:10 * 60000]
    for env in ({"SEQKIT_NO_LIBDEFLATE": "1"}, {"SEQKIT_NO_MMAP": "1"}):
        rc, out = run(env)
        assert rc == 3 and out == raw[:10 * 60000]
    f.write_bytes(gzip.compress(raw[:500_000]) + gzip.compress(raw[500_000:700_000]))    # gzip members without the BGZF field
    assert run() == (0, raw[:700_000])
    f.write_bytes(b"")
    assert run() == (0, b"")
    f.write_bytes(b"not gzip at all, just text\n" * 10)
    assert run()[0] == 3


def test_own_inflate_and_crc_against_zlib_under_asan(tmp_path):
    """host::inflate_raw / host::crc32_fast (what every BGZF block goes through when libdeflate is not there): thousands of
    zlib streams of every level and strategy decode to the same bytes, damaged streams never crash and are never accepted
    unless zlib accepts them with the same output, CRCs of random ranges equal zlib's.  tests/cpp/inflate_test.cpp."""
    import os
    import subprocess
    from seqkit_amd import build
    exe = tmp_path / "inflate_test"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I", build.CSRC, "-I", os.path.join(build.REPO, "include"), "-o", str(exe),
                    os.path.join(build.REPO, "tests", "cpp", "inflate_test.cpp"), os.path.join(build.CSRC, "host_inflate.cpp"), "-lz"], check=True)
    out = subprocess.run([str(exe), "800"], stdout=subprocess.PIPE, check=True, env={"ASAN_OPTIONS": "detect_leaks=0"}).stdout.decode()
    assert out.startswith("ok: 800 streams"), out
