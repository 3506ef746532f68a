"""CPU: the C++ `fasta` host against the oracle CLI for the parts that need no GPU (argument grammar, exit codes,
`fasta add barcode`, stream plumbing).  Anything arithmetic is GPU-only and lives in test_cli_gpu.py."""
import gzip

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
                    os.path.join(build.REPO, "tests", "cpp", "gz_writer_test.cpp"), os.path.join(csrc, "host_common.cpp"),
                    "-L", build.LIBDIR, "-lseqkit_hip", f"-Wl,-rpath,{build.LIBDIR}", "-lz"], check=True)
    for nf, total in ((5, 3_000_000), (2, 0), (40, 100_000)):
        d = tmp_path / f"o{nf}_{total}"
        d.mkdir()
        subprocess.run([str(exe), str(d), str(nf), str(total)], check=True, timeout=120)
        for i in range(nf):
            got = gzip.open(d / f"f{i}.gz", "rb").read()
            exp = bytes((ord("A") + (k * (i + 3)) % 23) for k in range(total)) if total <= 200_000 else None
            assert len(got) == total
            if exp is not None:
                assert got == exp
            else:
                import numpy as np
                k = np.arange(total, dtype=np.int64)
                assert np.array_equal(np.frombuffer(got, dtype=np.uint8), (65 + (k * (i + 3)) % 23).astype(np.uint8))
