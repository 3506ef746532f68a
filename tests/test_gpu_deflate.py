"""F2 on the device (src/common.rs:49-81): what sk_bgzf_deflate writes must be valid BGZF — every member a gzip
member with the BC subfield, the right CRC32 and ISIZE, a DEFLATE stream zlib accepts — and inflate to exactly the bytes that
went in (parity is on the decompressed stream: the reference's compressed bytes are its gzip child's).  Every kind of input the
three phases have a path for: nothing to match, everything a match, runs, matches at the longest distance a block allows,
blocks of 1 .. 5 bytes, exactly 0xff00 bytes, skewed histograms (codes that want more than 15 bits), bytes the host stores."""
import gzip
import zlib

import numpy as np
import pytest

from seqkit_amd import synth
from tests import bam_spec

pytestmark = pytest.mark.gpu


def corpus():
    rng = np.random.default_rng(23)
    seq, qual = synth.make_reads(3000, 150, seed=23)
    fastq = synth.fastq_text(seq, qual, prefix="SIM:23")
    out = [("fastq text, several blocks", fastq[:300_000]),
           ("random bytes (stored)", rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes()),
           ("zeros", bytes(0xff00)),
           ("one byte", b"A"), ("two", b"AB"), ("three", b"ABC"), ("four", b"ABCD"), ("five", b"AAAAA"),
           ("exactly a block", (b"ACGTTGCAAC" * 7000)[:0xff00]),
           ("a block and one byte", (b"ACGTTGCAAC" * 7000)[:0xff00 + 1]),
           ("period 3", b"abc" * 30000),
           ("4 letters uniform (no matches worth taking)", rng.integers(0, 4, 100_000, dtype=np.uint8).tobytes()),
           ("far matches", (lambda p: p + rng.integers(0, 256, 30000, dtype=np.uint8).tobytes() + p + rng.integers(0, 256, 32000, dtype=np.uint8).tobytes() + p)(rng.integers(0, 256, 500, dtype=np.uint8).tobytes())),
           ]
    # a Fibonacci-shaped histogram: an unbounded Huffman code would be deeper than 15 bits
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    skew = np.concatenate([np.full(f, i, dtype=np.uint8) for i, f in enumerate(fib)])
    rng.shuffle(skew)
    out.append(("fibonacci histogram", skew.tobytes()[:0xff00]))
    return out


@pytest.mark.parametrize("name,data", corpus(), ids=[c[0] for c in corpus()])
def test_deflate_members_are_valid_bgzf_and_inflate_to_the_input(ctx, name, data):
    comp = ctx.bgzf_deflate(data)
    blocks = list(bam_spec.bgzf_blocks(comp))                   # header, BC subfield, CRC32, ISIZE, zlib inflates the payload: asserted inside
    assert b"".join(blocks) == data
    assert all(len(b) <= 0xff00 for b in blocks) and len(blocks) == -(-len(data) // 0xff00)
    assert gzip.decompress(comp) == data
    if name.startswith("fastq"):
        ratio = len(data) / len(comp)
        assert ratio > 1.85, ratio                              # (zlib makes 1.82 of this text at level 1, 1.97 at level 6)
    if name.startswith(("zeros", "period")):
        assert len(comp) < len(data) // 50
    if name.startswith("random"):
        assert len(comp) <= len(data) + 31 * len(blocks)       # stored: 18 + 5 + 8 bytes a member


def test_deflate_many_blocks_of_every_size(ctx):
    rng = np.random.default_rng(29)
    words = [rng.integers(65, 91, int(rng.integers(2, 12)), dtype=np.uint8).tobytes() for _ in range(200)]
    for trial in range(6):
        n = int(rng.integers(1, 400_000))
        kind = trial % 3
        if kind == 0:
            data = b" ".join(words[int(j)] for j in rng.integers(0, 200, n // 5 + 1))[:n]
        elif kind == 1:
            data = bytes(rng.integers(0, 6, n // 7 + 1, dtype=np.uint8).repeat(7))[:n]
        else:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        block = int(rng.choice([0xff00, 1000, 77, 4097]))
        comp = ctx.bgzf_deflate(data, block=block)
        assert b"".join(bam_spec.bgzf_blocks(comp)) == data, (trial, n, block)
