"""Synthetic inputs of SURVEY.md §8(d): fixed-seed FASTQ-shaped batches, sample sheets, BAM cores.

Generators only (numpy on the host, torch on the device for the full-size bench shard); no
seqkit arithmetic lives here.  Seeds and distributions follow the config table of SURVEY.md §8(d).
"""
from __future__ import annotations

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def quality_mu(L: int) -> np.ndarray:
    p = np.arange(L, dtype=np.float64)
    return 36.0 - 16.0 * (p / max(L - 1, 1)) ** 2


def make_reads(n: int, L: int = 150, seed: int = 1, n_frac: float = 0.005):
    """seq[n, L], qual[n, L] (Phred+33): bases uniform ACGT with n_frac 'N'; q ~ clamp(round(N(mu(p), 6)), 2, 40)."""
    rng = np.random.default_rng(seed)
    seq = BASES[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    seq[rng.random((n, L)) < n_frac] = ord("N")
    q = np.rint(rng.normal(quality_mu(L)[None, :], 6.0, size=(n, L)))
    qual = (np.clip(q, 2, 40) + 33).astype(np.uint8)
    return np.ascontiguousarray(seq), np.ascontiguousarray(qual)


def add_forced_classes(qual: np.ndarray, seed: int = 2, frac: float = 0.05) -> np.ndarray:
    """cfg 2 parity classes, `frac` of the rows each: all '#', all 'I', '###..III', one 0x1F byte."""
    rng = np.random.default_rng(seed)
    n, L = qual.shape
    cls = rng.random(n)
    qual = qual.copy()
    qual[cls < frac] = ord("#")
    qual[(cls >= frac) & (cls < 2 * frac)] = ord("I")
    m = (cls >= 2 * frac) & (cls < 3 * frac)
    qual[m, : L // 2] = ord("#")
    qual[m, L // 2:] = ord("I")
    m = np.nonzero((cls >= 3 * frac) & (cls < 4 * frac))[0]
    qual[m, rng.integers(0, L, size=m.size)] = 0x1F
    return qual


def ragged_lengths(n: int, L: int, seed: int = 7) -> np.ndarray:
    """len[n] in 0..L with the edge lengths 0, 1, L forced in."""
    rng = np.random.default_rng(seed)
    ln = rng.integers(0, L + 1, size=n).astype(np.uint16)
    if n >= 4:
        ln[0], ln[1], ln[2], ln[-1] = 0, 1, L, 0
    return ln


def _distant_kmers(count: int, k: int, min_dist: int, rng) -> np.ndarray:
    out: list[np.ndarray] = []
    while len(out) < count:
        c = BASES[rng.integers(0, 4, size=k)]
        if all(int((c != o).sum()) >= min_dist for o in out):
            out.append(c)
    return np.stack(out)


def make_sheet(S: int, half: int = 8, dual: bool = False, seed: int = 3, min_dist: int = 3) -> np.ndarray:
    """Sample-sheet barcodes [S, L].  single: S k-mers pairwise >= min_dist apart.  dual: `i7+i5`
    (2*half+1 chars) built from distinct i7 x i5 combinations, pairwise >= min_dist within each half."""
    rng = np.random.default_rng(seed)
    if not dual:
        return np.ascontiguousarray(_distant_kmers(S, half, min_dist, rng))
    n7 = int(np.ceil(np.sqrt(S * 1.5)))
    n5 = int(np.ceil(S / n7))
    i7 = _distant_kmers(n7, half, min_dist, rng)
    i5 = _distant_kmers(n5, half, min_dist, rng)
    rows = []
    for s in range(S):
        rows.append(np.concatenate([i7[s % n7], np.frombuffer(b"+", dtype=np.uint8), i5[s // n7]]))
    return np.ascontiguousarray(np.stack(rows))


def observe_barcodes(table: np.ndarray, n: int, seed: int = 3, p_exact: float = 0.85, p_sub: float = 0.10,
                     halves: int = 1):
    """bc[n, L]: p_exact exact copies of a random sample's barcode, p_sub with one substitution (uniform
    position, uniform from ACGTN != original) per half, the rest uniform random k-mers.  The error mix is
    applied per half for dual-index sheets (the '+' separator is never touched)."""
    rng = np.random.default_rng(seed)
    S, L = table.shape
    truth = rng.integers(0, S, size=n)
    bc = table[truth].copy()
    alphabet = np.frombuffer(b"ACGTN", dtype=np.uint8)
    half = (L - (halves - 1)) // halves
    for h in range(halves):
        lo = h * (half + 1)
        u = rng.random(n)
        sub = np.nonzero((u >= p_exact) & (u < p_exact + p_sub))[0]
        pos = lo + rng.integers(0, half, size=sub.size)
        idx = rng.integers(0, 5, size=sub.size)
        same = alphabet[idx] == bc[sub, pos]
        idx[same] = (idx[same] + 1) % 5
        bc[sub, pos] = alphabet[idx]
        rnd = np.nonzero(u >= p_exact + p_sub)[0]
        bc[rnd, lo:lo + half] = BASES[rng.integers(0, 4, size=(rnd.size, half))]
    return np.ascontiguousarray(bc), truth


def make_bam_cores(n: int, seed: int = 5):
    """cfg 5: flag/tid/mtid/tlen columns with a realistic mix."""
    rng = np.random.default_rng(seed)
    flag = np.full(n, 0x1, dtype=np.uint16)
    first = (np.arange(n) % 2 == 0)
    flag |= np.where(first, 0x40, 0x80).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.03, 0x4, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.02, 0x8, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.08, 0x400, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.01, 0x100, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.005, 0x800, 0).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.5, 0x10, 0x20).astype(np.uint16)
    flag |= np.where(rng.random(n) < 0.9, 0x2, 0).astype(np.uint16)
    tid = rng.integers(0, 24, size=n).astype(np.int32)
    mtid = tid.copy()
    other = rng.random(n) >= 0.98
    mtid[other] = rng.integers(-1, 24, size=int(other.sum())).astype(np.int32)
    mag = np.rint(rng.lognormal(np.log(170.0), 0.35, size=n))
    big = rng.random(n) < 0.005
    mag[big] = rng.integers(5001, 2_000_000, size=int(big.sum()))
    tlen = (mag * np.where(rng.random(n) < 0.5, 1, -1)).astype(np.int32)
    tlen[rng.random(n) < 0.01] = 0
    return flag, tid, mtid, tlen


def write_bam_file(path: str, n_records: int, seed: int = 5, kind: str = "random", unit_records: int = 100_000, level: int = 1):
    """cfg 5 as a FILE: a BGZF / BAM file of n_records records (rounded down to whole units) whose core fields are make_bam_cores'
    mix, written the way htslib writes (a block is flushed rather than a record split).  kind "random": bases and qualities drawn
    uniformly (literals, about 1.5 : 1 — a worst case for an inflater); "sorted": reads drawn from a small genome in position order
    with binned qualities (what a coordinate-sorted BAM of a current instrument looks like, about 5 : 1).  ONE unit of
    unit_records records is built and compressed, then repeated.  Returns (records written, flag, tid, mtid, tlen of ONE unit,
    repeats): the expected reduction is the unit's, times repeats."""
    import struct
    import zlib
    rng = np.random.default_rng(seed)
    flag, tid, mtid, tlen = make_bam_cores(unit_records, seed=seed)
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)
    genome = codes[rng.integers(0, 4, size=300_000)]
    qbins = np.array([2, 12, 23, 37], dtype=np.uint8)
    unit = bytearray()
    ends = []
    for i in range(unit_records):
        name = b"A00123:45:HXXXXXXX:1:%d:%d:%d\0" % (1101 + i // 5000, 1000 + (i * 7) % 30000, 1000 + (i * 13) % 30000)
        if kind == "random":
            nib = codes[rng.integers(0, 4, size=150)]
            q = rng.integers(2, 41, size=150, dtype=np.uint8)
        else:
            p = (i * 3) % (len(genome) - 150)
            nib = genome[p:p + 150].copy()
            if i % 3 == 0:
                nib[(i * 7) % 150] = codes[i % 4]
            q = qbins[np.minimum(3, rng.geometric(0.75, size=150) - 1)][::-1].copy()
            q[:100] = 37
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        body = struct.pack("<iiBBHHHiiii", int(tid[i]), i * 3, len(name), 60, 4680, 1, int(flag[i]), 150, int(mtid[i]), i * 3 + 100, int(tlen[i])) + name
        body += struct.pack("<I", 150 << 4) + packed + q.tobytes()
        unit += struct.pack("<i", len(body)) + body
        ends.append(len(unit))
    unit = bytes(unit)

    def bgzf(data):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))

    blocks, lo, prev = [], 0, 0
    for e in ends:
        if e - lo > 0xff00:
            blocks.append(bgzf(unit[lo:prev]))
            lo = prev
        prev = e
    blocks.append(bgzf(unit[lo:]))
    body = b"".join(blocks)
    reps = max(1, n_records // unit_records)
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    hdr = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 24)
    for r in range(24):
        nm = b"chr%d\0" % (r + 1)
        hdr += struct.pack("<i", len(nm)) + nm + struct.pack("<i", 1 << 28)
    with open(path, "wb") as f:
        f.write(bgzf(hdr))
        for _ in range(reps):
            f.write(body)
        f.write(bgzf(b""))
    return reps * unit_records, flag, tid, mtid, tlen, reps


def fastq_text(seq: np.ndarray, qual: np.ndarray, prefix: str = "SIM:1", lengths=None, headers=None) -> bytes:
    """Four-line FASTQ text of a batch (header `@<prefix>:<i>` unless headers are given)."""
    out = []
    n = seq.shape[0]
    for i in range(n):
        l = seq.shape[1] if lengths is None else int(lengths[i])
        h = headers[i] if headers is not None else f"@{prefix}:{i}".encode()
        out.append(h + b"\n" + seq[i, :l].tobytes() + b"\n+\n" + qual[i, :l].tobytes() + b"\n")
    return b"".join(out)
