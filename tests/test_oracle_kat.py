"""CPU: the oracle against the hand-derived known-answer vectors.

The reference has no tests of its own for this path, so these vectors — derived
by hand from the cited source lines — are what pins the oracle.  Parity with the reference
binary itself remains unpinned.
"""
import subprocess

import numpy as np
import pytest


def b(s: str) -> bytes:
    return s.encode("latin-1")


def test_trim_kat(oracle, golden):
    g = golden["trim_by_quality"]
    for c in g["cases"]:
        assert oracle.trim_lowest_k(b(c["qual"]), g["min_baseq"]) == c["lowest_k"], c


def test_trim_kat_cli(oracle, golden, tmp_path):
    g = golden["trim_by_quality"]
    fq = tmp_path / "t.fq"
    fq.write_bytes(b"".join(b"@r%d\n" % i + b(c["seq"]) + b"\n+x\n" + b(c["qual"]) + b"\n" for i, c in enumerate(g["cases"])))
    out = subprocess.run([oracle.FASTA_BIN, "trim", "by", "quality", str(fq), str(g["min_baseq"])], stdout=subprocess.PIPE, check=True).stdout
    exp = b"".join(b"@r%d\n" % i + b(c["out_seq"]) + b"\n+\n" + b(c["out_qual"]) + b"\n" for i, c in enumerate(g["cases"]))
    assert out == exp


def test_mask_kat(oracle, golden, tmp_path):
    for c in golden["mask_by_quality"]["cases"]:
        assert oracle.mask_bytes(b(c["seq"]), b(c["qual"]), c["min_baseq"]) == b(c["out"]), c
        fq = tmp_path / "m.fq"
        fq.write_bytes(b"@r\n" + b(c["seq"]) + b"\n+\n" + b(c["qual"]) + b"\n")
        out = subprocess.run([oracle.FASTA_BIN, "mask", "by", "quality", str(fq), str(c["min_baseq"])], stdout=subprocess.PIPE, check=True).stdout
        assert out == b"@r\n" + b(c["out"]) + b"\n+\n" + b(c["qual"]) + b"\n"


def test_demux_kat(oracle, golden):
    g = golden["demultiplex"]
    for c in g["cases"]:
        table = np.array([list(b(x)) for x in c["sheet"]], dtype=np.uint8)
        bc = np.array([list(b(c["observed"]))], dtype=np.uint8)
        assign, low, first, last, counts = oracle.demux_batch(table, bc, g["max_diff"])
        assert (int(low[0]), int(first[0]), int(last[0]), int(assign[0])) == (c["lowest_diff"], c["first"], c["last"], c["assign"]), c
        S = len(c["sheet"])
        assert counts[S] == 1 and counts[S + 1] == (1 if c["assign"] >= 0 else 0) and counts[S + 2] == (1 if c["assign"] == -2 else 0)


def test_barcode_diff_wildcards(oracle):
    assert oracle.barcode_diff(b"ACGT", b"NNNN") == 0
    assert oracle.barcode_diff(b"ACGT", b"UUUU") == 0
    assert oracle.barcode_diff(b"NNNN", b"ACGT") == 4      # observed N is an ordinary byte
    assert oracle.barcode_diff(b"acgt", b"ACGT") == 4      # case-sensitive
    assert oracle.barcode_diff(b"AC+T", b"AC+T") == 0


def test_headers_kat(oracle, golden, tmp_path):
    for c in golden["headers"]["bc_field"]:
        h = b(c["header"])
        st, en = oracle.find_bc_field(h)
        assert h[st + 4:en] == b(c["barcode"])
        rest = h[:st] + h[en:]
        assert rest[:oracle.trim_end_len(rest)] == b(c["written"])
    for c in golden["headers"]["add_barcode"]:
        fq = tmp_path / "r.fq"
        ix = tmp_path / "i.fq"
        fq.write_bytes(b(c["header"]) + b"ACGT\n+\nIIII\n")
        ix.write_bytes(b"@i\n" + b(c["index_seq"]) + b"+\nIIII\n")
        out = subprocess.run([oracle.FASTA_BIN, "add", "barcode", str(fq), str(ix)], stdout=subprocess.PIPE, check=True).stdout
        assert out == b(c["out"]) + b"ACGT\n+\nIIII\n"


def test_bc_regex_semantics(oracle):
    assert oracle.find_bc_field(b"@r BC:XACGT BC:ACGT\n") == (11, 19)    # first hit needs >= 1 class byte
    assert oracle.find_bc_field(b"@r BC:\n") is None
    assert oracle.find_bc_field(b"@r BC:ACGT+TTGA:9\n") == (2, 15)
    assert oracle.find_bc_field(b"@rBC:ACGT\n") is None                   # the leading space is part of the pattern
    assert oracle.find_bc_field(b"@r BC:acgtn+N\n") == (2, 13)


def test_rust_trim_end(oracle):
    assert oracle.trim_end_len(b"abc \t\r\n") == 3
    assert oracle.trim_end_len(b"abc\x0b\x0c") == 3
    assert oracle.trim_end_len(b"abc\x1f") == 4               # 0x1C..0x1F are not White_Space (Python strip() differs)
    assert oracle.trim_end_len(b"abc\x1c\n") == 4
    assert oracle.trim_end_len("abc  　".encode()) == 3
    assert oracle.trim_end_len("abc​".encode()) == 6     # zero-width space is not White_Space
    assert oracle.trim_end_len(b"") == 0 and oracle.trim_end_len(b" \n") == 0


def test_utf8_validation(oracle):
    assert oracle.utf8_valid(b"ACGT\n") and oracle.utf8_valid("é€😀".encode())
    for bad in (b"\xff", b"\xc0\xaf", b"\xe0\x80\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xe2\x82", b"\x80"):
        assert not oracle.utf8_valid(bad), bad


def test_bam_kat(oracle, golden):
    g = golden["bam"]
    flag = np.array(g["flags"], dtype=np.uint16)
    tid = np.array(g["tid"], dtype=np.int32)
    mtid = np.array(g["mtid"], dtype=np.int32)
    tlen = np.array(g["tlen"], dtype=np.int32)
    counters, hist, total = oracle.bam_flag_tlen(flag, tid, mtid, tlen, 5000)
    assert list(map(int, counters)) == [g["total"], g["aligned"], g["duplicate"]]
    nz = {str(i): int(hist[i]) for i in np.nonzero(hist)[0]}
    assert nz == g["hist_nonzero"] and total == 1


def bam_sequence_case(c, stride=8):
    raw = bytes.fromhex(c["seq4_hex"])
    seq4 = np.zeros((1, stride // 2), dtype=np.uint8)
    seq4[0, :len(raw)] = np.frombuffer(raw, dtype=np.uint8)
    qual = np.zeros((1, stride), dtype=np.uint8)
    qual[0, :len(c["qual"])] = c["qual"]
    return seq4, qual, np.array([len(c["qual"])], dtype=np.uint16), np.array([c["flag"]], dtype=np.uint16)


def test_bam_sequence_kat(oracle, golden):
    g = golden["bam_sequence"]
    for c in g["cases"]:
        seq4, qual, ln, flag = bam_sequence_case(c)
        out = oracle.bam_sequence_batch(seq4, qual, ln, flag, g["min_baseq"])
        assert out[0, :ln[0]].tobytes() == c["out"].encode(), c


def test_census_kat(oracle, golden, tmp_path):
    g = golden["census"]
    rows = []
    for h, want in zip(g["headers"], g["barcodes"]):
        m = oracle.find_bc_field_stats(h.encode())
        got = None if m is None else h.encode()[m[0] + 4:m[1]].decode()
        assert got == want, h
        if got is not None:
            rows.append(got.encode())
    bc = np.zeros((len(rows), 8), dtype=np.uint8)
    for i, r in enumerate(rows):
        bc[i, :len(r)] = np.frombuffer(r, dtype=np.uint8)
    assert [[k.decode(), c] for k, c, _ in oracle.census(bc)] == g["counts_first_seen"]


@pytest.mark.parametrize("m", [0, 2, 20, 30, 41, 223, 224, 255])
def test_trim_closed_form(oracle, m):
    """The suffix-sum closed form of SURVEY.md §8(a) T1 (what the GPU kernel computes) equals the loop."""
    rng = np.random.default_rng(100 + m)
    for _ in range(300):
        n = int(rng.integers(0, 40))
        q = rng.integers(0, 256, size=n, dtype=np.uint8) if rng.random() < 0.3 else rng.integers(33, 75, size=n).astype(np.uint8)
        v = ((q.astype(np.int64) - 33) % 256) - m
        S = -50 + np.cumsum(v[::-1])[::-1] if n else np.zeros(0)
        pos = np.nonzero(S > 0)[0]
        bnd = pos.max() if pos.size else -1
        k = n
        if bnd + 1 < n:
            seg = S[bnd + 1:]
            if seg.min() < -50:
                k = bnd + 1 + int(np.nonzero(seg == seg.min())[0].max())
        assert oracle.trim_lowest_k(q.tobytes(), m) == k
