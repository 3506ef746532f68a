"""CPU: the per-read TEXT commands of the `fasta` host against the oracle CLI — same stdout, stderr
and exit code.  They are line filters with no device work, so they run without a GPU."""
import numpy as np
import pytest

from seqkit_amd import synth
from tests import cli_util as cu


@pytest.fixture(scope="module")
def bins(hip_lib, oracle):
    from seqkit_amd import build
    build.build_hosts()
    return {"fasta": (cu.FASTA, oracle.FASTA_BIN)}


def both(bins, tool, args, tmp_path, stdin=None, same_stderr=True):
    a = cu.run(bins[tool][0], args, cwd=tmp_path, stdin=stdin)
    b = cu.run(bins[tool][1], args, cwd=tmp_path, stdin=stdin)
    assert a[0] == b[0], (a[0], b[0], a[2][-500:], b[2][-500:])
    assert a[1] == b[1]
    if same_stderr:
        assert a[2] == b[2]
    return a, b


def mixed_text(n, seed, fasta_every=0, umi=True):
    """FASTQ (and, every `fasta_every`-th record, FASTA) records with Illumina-style headers, some with UMI: tags."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        ln = int(rng.integers(0, 60))
        seq = synth.BASES[rng.integers(0, 4, size=ln)].tobytes()
        qual = bytes(rng.integers(33, 74, size=ln, dtype=np.uint8))
        h = b"M01:23:FC:1:%d:%d:%d" % (i % 7, i, int(rng.integers(0, 99999)))
        if umi and i % 3 == 0:
            h += b" UMI:" + synth.BASES[rng.integers(0, 4, size=8)].tobytes() + b" tail"
        h += b" 1:N:0:" + synth.BASES[rng.integers(0, 4, size=8)].tobytes()
        if fasta_every and i % fasta_every == 0:
            out.append(b">" + h + b"\n" + seq + b"\n")
        else:
            out.append(b"@" + h + b"\n" + seq + b"\n+" + (h if i % 5 == 0 else b"") + b"\n" + qual + b"\n")
    return b"".join(out)


def test_fasta_trim_fixed_cli(bins, tmp_path):
    fq = tmp_path / "t.fq"
    fq.write_bytes(mixed_text(2000, seed=1, fasta_every=4))
    for args in ([], ["--first=3"], ["--last", "5"], ["--first=10", "--last=10"], ["--first=100"], ["--fi=1", "--la=1"], ["--first=+2"]):
        both(bins, "fasta", ["trim"] + args + [str(fq)], tmp_path)
    both(bins, "fasta", ["trim", "--first=3", "-"], tmp_path, stdin=fq.read_bytes())
    # quirks: CRLF (trim_end drops the \r for the length only), quality shorter than the bases (slice panic after the
    # header and bases are out), multi-byte characters cut in the middle, record cut off by end of file
    fq.write_bytes(b"@a\nACGTACGT\r\n+\nIIIIIIII\r\n")
    both(bins, "fasta", ["trim", "--first=1", "--last=1", str(fq)], tmp_path)
    fq.write_bytes(b"@a\nACGTACGT\n+\nIII\n@b\nAC\n+\nII\n")
    a, *_ = both(bins, "fasta", ["trim", "--first=1", str(fq)], tmp_path, same_stderr=False)
    assert a[0] == 101 and a[1] == b"@a\nCGTACGT\n"
    fq.write_bytes("@a\nA\u00e9GT\n+\nIIII\n".encode())
    a, *_ = both(bins, "fasta", ["trim", "--first=2", str(fq)], tmp_path, same_stderr=False)
    assert a[0] == 101 and a[1] == b""
    fq.write_bytes(b"@a\nACGT")
    both(bins, "fasta", ["trim", "--last=1", str(fq)], tmp_path)
    fq.write_bytes(b"@a\nACGT\n+\nIIII\nbad\n")
    both(bins, "fasta", ["trim", str(fq)], tmp_path)
    for args in (["trim"], ["trim", "--first=x", str(fq)], ["trim", "--last=-1", str(fq)], ["trim", "--first=1", "missing.fq"], ["trim", "--first=x", "missing.fq"],
                 ["trim", str(fq), "extra"], ["trim", "--first=18446744073709551616", str(fq)]):
        both(bins, "fasta", args, tmp_path)
    fq.write_bytes(b"@a\nACGT\n+\nIIII\n")
    both(bins, "fasta", ["trim", "--first=18446744073709551615", "--last=3", str(fq)], tmp_path, same_stderr=False)   # usize addition wraps


def test_fasta_extract_dual_umi_cli(bins, tmp_path):
    a = mixed_text(1000, seed=2, umi=False)
    recs = a.split(b"\n@M01")
    inter = tmp_path / "i.fq"
    inter.write_bytes(a)                                                     # consecutive records play the two mates
    for args in ([], ["--first-bases=0"], ["--first-bases", "3"], ["--first=5"]):
        both(bins, "fasta", ["extract", "dual", "umi"] + args + [str(inter)], tmp_path, same_stderr=False)
    inter.write_bytes(b">a x \nACGTAC\n>a y\nTTGCAA\n>b\nAC\n>b\nGT\n")
    x, *_ = both(bins, "fasta", ["extract", "dual", "umi", "--first-bases=2", str(inter)], tmp_path)
    assert x[1] == b">a x RX:AC+TT\nGTAC\n>a y RX:AC+TT\nGCAA\n>b RX:AC+GT\n\n>b RX:AC+GT\n\n"
    inter.write_bytes(b"@a\nACGTAC\n+\nIIIIII\n@a\nTTGCAA\n+\nJJJJJJ\n")
    x, *_ = both(bins, "fasta", ["extract", "dual", "umi", "--first-bases=4", str(inter)], tmp_path)
    assert x[1] == b"@a RX:ACGT+TTGC\nAC\n+\nII\n@a RX:ACGT+TTGC\nAA\n+\nJJ\n"
    x, *_ = both(bins, "fasta", ["extract", "dual", "umi", "--first-bases=8", str(inter)], tmp_path, same_stderr=False)
    assert x[0] == 101 and x[1] == b""                                       # seq_1[0..8] is out of range
    inter.write_bytes(b"@a\nACGTAC\n+\nIIIIII\n>a\nTTGCAA\n")
    both(bins, "fasta", ["extract", "dual", "umi", str(inter)], tmp_path)     # Invalid FASTQ record
    inter.write_bytes(b">a\nACGTAC\n")
    both(bins, "fasta", ["extract", "dual", "umi", str(inter)], tmp_path)     # mate missing
    inter.write_bytes(b"a\nACGTAC\n")
    both(bins, "fasta", ["extract", "dual", "umi", str(inter)], tmp_path)
    for args in (["extract", "dual", "umi"], ["extract", "dual", "umi", "--first-bases=z", str(inter)], ["extract", "dual", "umi", "nofile"]):
        both(bins, "fasta", args, tmp_path)


def test_fasta_convert_basespace_simplify_interleave_cli(bins, tmp_path):
    fq = tmp_path / "b.fq"
    fq.write_bytes(mixed_text(1500, seed=3, fasta_every=6))
    both(bins, "fasta", ["convert", "basespace", str(fq)], tmp_path)
    both(bins, "fasta", ["simplify", "read", "ids", str(fq)], tmp_path)
    both(bins, "fasta", ["simplify", "read", "ids", "--discard-umi", str(fq)], tmp_path)
    both(bins, "fasta", ["simplify", "read", "ids", "--alphanumeric", "--disc", str(fq)], tmp_path)
    fq2 = tmp_path / "b2.fq"
    fq2.write_bytes(mixed_text(1500, seed=3, fasta_every=6).replace(b" 1:N:0:", b" 2:N:0:"))
    a, *_ = both(bins, "fasta", ["interleave", str(fq), str(fq2)], tmp_path)
    assert a[1].count(b"\n") == 2 * fq.read_bytes().count(b"\n")
    # quirks
    fq.write_bytes(b"@nocolon\nAC\n+\nII\n@x:y: \nAC\n+\nII\n>fa:TTTT  \nACGT\nbad:AAAA\nAC\n")
    a, *_ = both(bins, "fasta", ["convert", "basespace", str(fq)], tmp_path)
    assert a[0] == 255 and a[1] == b"@1 BC:@nocolon\nAC\n+\nII\n@2\nAC\n+\nII\n@3 BC:TTTT\nACGT\n@4 BC:AAAA\n"
    fq.write_bytes("@r UMI:AC\u00a0GT x\nAC\n+x\nII\n>r2 UMI:\nAC\n@r3 UMI:\u00e9\u00e9 UMI:zz\nAC\n+\nII\n".encode())
    a, *_ = both(bins, "fasta", ["simplify", "read", "ids", str(fq)], tmp_path)
    assert a[1] == "@1 UMI:AC\nAC\n+\nII\n>2 UMI:\nAC\n@3 UMI:\u00e9\u00e9\nAC\n+\nII\n".encode()
    fq.write_bytes(b"@r\nAC\n+\nII\n\n")
    both(bins, "fasta", ["simplify", "read", "ids", str(fq)], tmp_path)       # a blank line is not a header
    fq.write_bytes(b"@r\nAC")
    both(bins, "fasta", ["simplify", "read", "ids", str(fq)], tmp_path)
    fq.write_bytes(b"@a\nAC\n+\nII\n>b\nGG\n")
    fq2.write_bytes(b"@a\nTT\n+\nJJ\n@b\nGG\n+\nII\n")
    a, *_ = both(bins, "fasta", ["interleave", str(fq), str(fq2)], tmp_path)
    assert a[0] == 255 and a[1] == b"@a\nAC\n+\nII\n@a\nTT\n+\nJJ\n>b\nGG\n"
    fq2.write_bytes(b"@a\nTT\n+\nJJ\n")
    both(bins, "fasta", ["interleave", str(fq), str(fq2)], tmp_path)          # second file runs out
    fq.write_bytes(b"x\n")
    both(bins, "fasta", ["interleave", str(fq), str(fq2)], tmp_path)
    for args in (["interleave", str(fq)], ["interleave", str(fq), "missing.fq"], ["convert", "basespace"], ["convert", "basespace", "missing.fq"],
                 ["simplify", "read", "ids"], ["simplify", "read", "ids", "--nope", str(fq)], ["simplify", "read"]):
        both(bins, "fasta", args, tmp_path)


def test_fasta_check_to_raw_and_base_qualities_cli(bins, tmp_path):
    fq = tmp_path / "c.fq"
    text = mixed_text(1200, seed=4, fasta_every=5)
    fq.write_bytes(text)
    a, _ = both(bins, "fasta", ["check", str(fq)], tmp_path)
    assert a[0] == 0 and a[1] == b"" and a[2] == b""
    both(bins, "fasta", ["to", "raw", str(fq)], tmp_path)
    lines = text.split(b"\n")
    fq.write_bytes(b"\n".join(lines[:40] + [b"oops"] + lines[40:]))          # a stray line: reported with the 10 lines before it
    a, _ = both(bins, "fasta", ["check", str(fq)], tmp_path)
    assert a[0] == 255 and a[2].startswith(b"ERROR: Missing ") and a[2].count(b"\n") >= 20
    fq.write_bytes(b"@r\nACGT\nIIII\n+\n")
    a, _ = both(bins, "fasta", ["check", str(fq)], tmp_path)
    assert a[2] == b"ERROR: Missing quality header prefix '+' on line 3:\n@r\n\nACGT\n\nIIII\n\n\n\n"
    fq.write_bytes(b"@r\nACGT\n")
    both(bins, "fasta", ["check", str(fq)], tmp_path)                         # '+' line missing at end of file
    fq.write_bytes(b"@r\nACGT\n+\nIIII\nx\n")
    both(bins, "fasta", ["to", "raw", str(fq)], tmp_path)
    # FASTA -> FASTQ -> FASTA
    fa = tmp_path / "c.fa"
    fa.write_bytes(b">s1 d\nACGTACGT\n>s2\n\n>s3\nAC")                     # last line without newline: one quality short
    a, _ = both(bins, "fasta", ["add", "base", "qualities", str(fa), "30"], tmp_path)
    assert a[1] == b"@s1 d\nACGTACGT\n+\n????????\n@s2\n\n+\n\n@s3\nAC+\n?\n"
    both(bins, "fasta", ["add", "base", "qualities", str(fa), "0"], tmp_path)
    both(bins, "fasta", ["add", "base", "qualities", str(fa), "94"], tmp_path)
    a, _ = both(bins, "fasta", ["add", "base", "qualities", str(fa), "95"], tmp_path, same_stderr=False)     # 33 + 95 is not ASCII: from_utf8().unwrap()
    assert a[0] == 101 and a[1] == b"@s1 d\nACGTACGT\n"
    both(bins, "fasta", ["add", "base", "qualities", str(fa), "250"], tmp_path)                              # wraps to 27
    for q in ("256", "-1", "x", ""):
        both(bins, "fasta", ["add", "base", "qualities", str(fa), q], tmp_path)
    fa.write_bytes(b">s1\n")
    a, _ = both(bins, "fasta", ["add", "base", "qualities", str(fa), "30"], tmp_path, same_stderr=False)     # no sequence line at all
    assert a[0] == 101 and a[1] == b"@s1\n"
    fa.write_bytes(b"@s1\nAC\n")
    both(bins, "fasta", ["add", "base", "qualities", str(fa), "30"], tmp_path)
    fq.write_bytes(text)
    a, _ = both(bins, "fasta", ["remove", "base", "qualities", str(fq)], tmp_path)                           # stops at the first FASTA record
    assert a[0] == 255 and a[1].startswith(b"")
    fq.write_bytes(mixed_text(500, seed=5))
    a, _ = both(bins, "fasta", ["remove", "base", "qualities", str(fq)], tmp_path)
    assert a[0] == 0 and a[1].count(b"\n") == 1000
    for args in (["check"], ["check", "a", "b"], ["check", "missing"], ["to", "raw"], ["to", "raw", "missing"], ["add", "base", "qualities", str(fa)],
                 ["add", "base", "qualities", "missing", "x"], ["remove", "base", "qualities"], ["remove", "base", "qualities", "missing"]):
        both(bins, "fasta", args, tmp_path)


def test_fasta_deinterleave_and_split_into_anchors_cli(bins, tmp_path):
    import gzip
    fq = tmp_path / "d.fq"
    fq.write_bytes(mixed_text(1000, seed=6))
    outs = {}
    for k, d in enumerate(("hip", "orc")):
        (tmp_path / d).mkdir()
        r = cu.run(bins["fasta"][k], ["deinterleave", str(fq), "out"], cwd=tmp_path / d)
        assert r[0] == 0 and r[1] == b"" and r[2] == b""
        outs[d] = cu.gunzip_dir(tmp_path / d)
    assert outs["hip"] == outs["orc"] and sorted(outs["hip"]) == ["out_1.fq.gz", "out_2.fq.gz"]
    assert outs["hip"]["out_1.fq.gz"].count(b"\n") == outs["hip"]["out_2.fq.gz"].count(b"\n") == 2000
    # interleave(deinterleave(x)) == x
    for k in (1, 2):
        (tmp_path / f"m{k}.fq").write_bytes(outs["hip"][f"out_{k}.fq.gz"])
    a, _ = both(bins, "fasta", ["interleave", str(tmp_path / "m1.fq"), str(tmp_path / "m2.fq")], tmp_path)
    assert a[1] == fq.read_bytes()
    fq.write_bytes(b"@a\nAC\n+\nII\n>b\nAC\n")
    for k, d in enumerate(("hip", "orc")):
        r = cu.run(bins["fasta"][k], ["deinterleave", str(fq), "bad"], cwd=tmp_path / d)
        assert r[0] == 255 and r[2] == b"ERROR: Interleaved FASTA records are not in consistent format.\n"
        assert gzip.open(tmp_path / d / "bad_1.fq.gz").read() == b"@a\nAC\n+\nII\n" and gzip.open(tmp_path / d / "bad_2.fq.gz").read() == b""
    for args in (["deinterleave", str(fq)], ["deinterleave", "missing", "p"]):
        both(bins, "fasta", args, tmp_path)

    fq.write_bytes(mixed_text(1500, seed=7, fasta_every=4))
    for n in ("0", "5", "20", "31"):
        both(bins, "fasta", ["split", "into", "anchors", str(fq), n], tmp_path)
    fq.write_bytes(b"@a\nACGTACGTAC\n+\nABCDEFGHIJ\n>b\nTTTTGGGGCC\n@short\nACG\n+\nIII\n>c\nAAAACCCC\n")
    a, _ = both(bins, "fasta", ["split", "into", "anchors", str(fq), "3"], tmp_path)
    # the skipped short record leaves its '+' and quality lines unread: they are taken for the next header and bases
    assert a[1].startswith(b"@1\nACG\n+\nABC\n@1\nTAC\n+\nHIJ\n>2\nTTT\n>2\nGCC\n")
    fq.write_bytes(b"@a\nACGTACGT\n+\nIII\n")
    a, _ = both(bins, "fasta", ["split", "into", "anchors", str(fq), "4"], tmp_path, same_stderr=False)      # quality shorter than the anchor
    assert a[0] == 101 and a[1] == b"@1\nACGT\n+\nIII\n\n"         # the first anchor's slice still fits ("III\\n"); the second panics
    fq.write_bytes(b"xa\nACGTACGT\n")
    both(bins, "fasta", ["split", "into", "anchors", str(fq), "2"], tmp_path)
    for args in (["split", "into", "anchors", str(fq)], ["split", "into", "anchors", str(fq), "x"], ["split", "into", "anchors", "missing", "3"],
                 ["split", "into", "anchors", str(fq), "9223372036854775808"]):
        both(bins, "fasta", args, tmp_path, same_stderr=False)


# ---- a third statement: the commands in a few lines of Python each, against the host binary (regular inputs only) -------
RUST_WS = "".join(chr(c) for c in [9, 10, 11, 12, 13, 32, 0x85, 0xA0, 0x1680, *range(0x2000, 0x200B), 0x2028, 0x2029, 0x202F, 0x205F, 0x3000])


def rust_trim_end(s: str) -> str:
    return s.rstrip(RUST_WS)                                              # str::trim_end: Unicode White_Space only (not 0x1c..0x1f)


def records(text: bytes):
    """[(header, seq, plus, qual)] with line ends kept; FASTA records have plus = qual = None."""
    lines = text.decode().splitlines(keepends=True)
    out, i = [], 0
    while i < len(lines):
        if lines[i].startswith("@"):
            out.append(tuple(lines[i:i + 4]))
            i += 4
        else:
            out.append((lines[i], lines[i + 1], None, None))
            i += 2
    return out


def test_text_commands_against_python_models(bins, tmp_path):
    import re
    text = mixed_text(800, seed=11, fasta_every=5)
    fq = tmp_path / "m.fq"
    fq.write_bytes(text)
    recs = records(text)
    run = lambda args: cu.run(bins["fasta"][0], args, cwd=tmp_path)

    # fasta trim --first=3 --last=4  (src/fasta_trim.rs:24-47)
    exp = []
    for h, s, p, q in recs:
        n = len(rust_trim_end(s))
        keep = 3 + 4 < n
        exp.append(h + (s[3:n - 4] if keep else "") + "\n")
        if p is not None:
            exp.append("+\n" + (q[3:n - 4] if keep else "") + "\n")
    rc, out, _ = run(["trim", "--first=3", "--last=4", str(fq)])
    assert rc == 0 and out.decode() == "".join(exp)

    # fasta to raw  (src/fasta_to_raw.rs:14-27)
    rc, out, _ = run(["to", "raw", str(fq)])
    assert rc == 0 and out.decode() == "".join(s for _, s, _, _ in recs)

    # fasta convert basespace  (src/fasta_convert_basespace.rs:25-45)
    exp = []
    for k, (h, s, p, q) in enumerate(recs, 1):
        bc = rust_trim_end(h).split(":")[-1]
        exp.append(f"@{k}" + (f" BC:{bc}" if bc else "") + "\n" + s + (p + q if p is not None else ""))
    rc, out, _ = run(["convert", "basespace", str(fq)])
    assert rc == 0 and out.decode() == "".join(exp)

    # fasta simplify read ids [--discard-umi]  (src/fasta_simplify_read_ids.rs:30-60)
    for discard in (False, True):
        exp = []
        for k, (h, s, p, q) in enumerate(recs, 1):
            m = None if discard else re.search(r" UMI:[^\s]*", h)
            exp.append(f"{h[0]}{k}" + (m.group(0) if m else "") + "\n" + s + ("+\n" + q if p is not None else ""))
        rc, out, _ = run(["simplify", "read", "ids"] + (["--discard-umi"] if discard else []) + [str(fq)])
        assert rc == 0 and out.decode() == "".join(exp)

    # fasta split into anchors <fastq> 7 on FASTA records and on FASTQ records that are long enough (:22-44)
    ok = [r for r in recs if len(rust_trim_end(r[1])) >= 14]
    a_in = tmp_path / "anch.fq"
    a_in.write_bytes("".join("".join(x for x in r if x is not None) for r in ok).encode())
    exp = []
    for k, (h, s, p, q) in enumerate(ok, 1):
        n = len(rust_trim_end(s))
        if p is not None:
            exp.append(f"@{k}\n{s[:7]}\n+\n{q[:7]}\n@{k}\n{s[n - 7:n]}\n+\n{q[n - 7:n]}\n")
        else:
            exp.append(f">{k}\n{s[:7]}\n>{k}\n{s[n - 7:n]}\n")
    rc, out, _ = run(["split", "into", "anchors", str(a_in), "7"])
    assert rc == 0 and out.decode() == "".join(exp) and len(ok) > 300

    # FASTQ only: remove base qualities, extract dual umi, interleave / deinterleave
    fqs = [r for r in recs if r[2] is not None]
    fqs = fqs[:len(fqs) // 2 * 2]
    only = tmp_path / "only.fq"
    only.write_bytes("".join("".join(r) for r in fqs).encode())
    rc, out, _ = run(["remove", "base", "qualities", str(only)])
    assert rc == 0 and out.decode() == "".join(">" + h[1:] + s for h, s, _, _ in fqs)
    long_enough = [(a, b) for a, b in zip(fqs[0::2], fqs[1::2]) if len(a[1]) > 5 and len(b[1]) > 5 and len(a[3]) > 5 and len(b[3]) > 5]
    umi_in = tmp_path / "umi.fq"
    umi_in.write_bytes("".join("".join(a) + "".join(b) for a, b in long_enough).encode())
    exp = []
    for a, b in long_enough:                                               # src/fasta_extract_dual_umi.rs:57-66
        umi = a[1][:5] + "+" + b[1][:5]
        for h, s, _, q in (a, b):
            exp.append(f"{rust_trim_end(h)} RX:{umi}\n{s[5:]}+\n{q[5:]}")
    rc, out, _ = run(["extract", "dual", "umi", "--first-bases=5", str(umi_in)])
    assert rc == 0 and out.decode() == "".join(exp) and len(long_enough) > 200
    m1, m2 = tmp_path / "m1.fq", tmp_path / "m2.fq"
    m1.write_bytes("".join("".join(r) for r in fqs[0::2]).encode())
    m2.write_bytes("".join("".join(r) for r in fqs[1::2]).encode())
    rc, out, _ = run(["interleave", str(m1), str(m2)])
    assert rc == 0 and out == only.read_bytes()

    # fasta add base qualities <fasta> 30 on the FASTA records (src/fasta_add_base_qualities.rs:19-26)
    fas = [r for r in recs if r[2] is None]
    fa = tmp_path / "only.fa"
    fa.write_bytes("".join(h + s for h, s, _, _ in fas).encode())
    rc, out, _ = run(["add", "base", "qualities", str(fa), "30"])
    assert rc == 0 and out.decode() == "".join("@" + h[1:] + s + "+\n" + "?" * (len(s) - 1) + "\n" for h, s, _, _ in fas)
