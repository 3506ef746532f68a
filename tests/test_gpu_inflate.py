"""B1 on the device (src/common.rs:121-157): BGZF blocks inflated by the GPU must be, byte for byte, what zlib
makes of them — every kind of DEFLATE block (stored, fixed, dynamic; several per BGZF block), codes longer than the decoding
tables' index bits, matches at every distance (1: a run; beyond the kernel's LDS ring; the full 32 KiB), blocks from empty to
64 KiB at every alignment of their bytes in the file and of their output in the stream.  Damaged blocks must be REPORTED
(status != 0), never crash or hang the device, and never decide anything: the caller's zlib does.  Then the record walk
over the inflated stream — blocks cut at record boundaries (htslib) and anywhere — against a plain Python walk, and the
whole-file entry point against the SoA kernel and the specification's reader (tests/bam_spec.py)."""
import os
import struct
import zlib

import numpy as np
import pytest

from tests import bam_spec
from tests.cli_util import write_bam

pytestmark = pytest.mark.gpu


def deflate_raw(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None, mem_level=8) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
    if flush_every is None:
        return c.compress(data) + c.flush()
    out = b""
    for o in range(0, len(data), flush_every):
        out += c.compress(data[o:o + flush_every]) + c.flush(zlib.Z_FULL_FLUSH)      # an empty stored block between deflate blocks
    return out + c.flush()


class Dev:
    """device buffers of one test, freed at the end"""

    def __init__(self, ctx):
        self.ctx, self.ptrs = ctx, []

    def put(self, arr: np.ndarray, pad=64) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.ctx.malloc_device(arr.nbytes + pad)
        self.ptrs.append(p)
        if arr.nbytes:
            self.ctx.copy_h2d(p, arr)
        self.ctx.sync()
        return p

    def empty(self, nbytes) -> int:
        p = self.ctx.malloc_device(nbytes + 64)
        self.ptrs.append(p)
        return p

    def get(self, p: int, n: int, dtype=np.uint8) -> np.ndarray:
        out = np.empty(n, dtype=dtype)
        if out.nbytes:
            self.ctx.copy_d2h(out, p)
        self.ctx.sync()
        return out

    def close(self):
        for p in self.ptrs:
            self.ctx.free_device(p)


def inflate_on_device(ctx, payloads, raws, gaps=None, out_gaps=None, crcs=None, check_crc=True):
    """payloads[i]: raw DEFLATE of raws[i].  They are laid out in one buffer with gaps[i] bytes before payload i (any alignment), their
    outputs in one stream with out_gaps[i] bytes before output i.  Returns (status u32[n], outputs [bytes])."""
    n = len(payloads)
    gaps = gaps or [0] * n
    out_gaps = out_gaps or [0] * n
    comp = bytearray()
    blocks = np.zeros(n, dtype=ctx.BGZF_BLOCK_DTYPE)
    out_off = 0
    for i, (p, r) in enumerate(zip(payloads, raws)):
        comp += bytes([0xA5]) * gaps[i]
        out_off += out_gaps[i]
        blocks[i] = (len(comp), len(p), len(r), out_off, (zlib.crc32(r) & 0xFFFFFFFF) if crcs is None else crcs[i], 0)
        comp += p
        out_off += len(r)
    comp += bytes(16)
    d = Dev(ctx)
    try:
        d_comp = d.put(np.frombuffer(bytes(comp), dtype=np.uint8))
        d_blocks = d.put(blocks.view(np.uint8))
        d_out = d.put(np.full(out_off + 16, 0xEE, dtype=np.uint8))
        d_status = d.put(np.full(n, 0xFFFFFFFF, dtype=np.uint32).view(np.uint8))
        ctx.bgzf_inflate_dev(d_comp, d_blocks, n, d_out, d_status, check_crc)
        ctx.sync()
        status = d.get(d_status, n, np.uint32)
        out = d.get(d_out, out_off + 16).tobytes()
    finally:
        d.close()
    outs = [out[int(b["out_off"]):int(b["out_off"]) + int(b["out_len"])] for b in blocks]
    # nothing outside the blocks' own ranges was written
    mask = np.ones(out_off + 16, dtype=bool)
    for b in blocks:
        mask[int(b["out_off"]):int(b["out_off"]) + int(b["out_len"])] = False
    assert (np.frombuffer(out, dtype=np.uint8)[mask] == 0xEE).all(), "bytes outside the blocks' outputs were written"
    return status, outs


def corpus(rng):
    """(name, raw bytes, deflate kwargs): every kind of block the decoder has a path for"""
    out = []
    text = (b"@SIM:1:%d 1:N:0 BC:ACGTACGT+TTGCAAGG\n" * 40) % tuple(range(40))
    recs = b"".join(struct.pack("<iiiBBHHHiiii", 200 + (i % 7), 0, i * 3, 8, 60, 4680, 1, 99 if i % 2 == 0 else 147, 150, 0, i * 3 + 170, 170 if i % 2 == 0 else -170)
                    + b"read%03d\0" % (i % 1000) + struct.pack("<I", 150 << 4) + rng.integers(0, 256, 75, dtype=np.uint8).tobytes()
                    + rng.integers(2, 41, 150, dtype=np.uint8).tobytes() for i in range(220))
    out.append(("empty", b"", {}))
    out.append(("one byte", b"A", {}))
    out.append(("two bytes", b"AB", {}))
    out.append(("text", text, {}))
    out.append(("bam-like 64k", recs[:65280], {"level": 1}))
    out.append(("bam-like level 9", recs[:60000], {"level": 9}))
    out.append(("zeros 64 KiB: runs (distance 1, length 258)", bytes(65536), {}))
    out.append(("period 3", b"abc" * 20000, {}))
    out.append(("period 300", rng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 200, {}))
    out.append(("random 64 KiB at level 6 (stored or literal-only)", rng.integers(0, 256, 65536, dtype=np.uint8).tobytes(), {}))
    out.append(("random at level 0: stored blocks", rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(), {"level": 0}))
    out.append(("fixed Huffman", text * 3, {"strategy": zlib.Z_FIXED}))
    out.append(("fixed Huffman, tiny", b"hello hello hello", {"strategy": zlib.Z_FIXED}))
    out.append(("huffman only (no matches)", rng.integers(0, 4, 30000, dtype=np.uint8).tobytes(), {"strategy": zlib.Z_HUFFMAN_ONLY}))
    out.append(("rle strategy", bytes(rng.integers(0, 3, 3000, dtype=np.uint8).repeat(11)), {"strategy": zlib.Z_RLE}))
    out.append(("several deflate blocks with empty stored blocks between", recs[:50000], {"flush_every": 7001}))
    out.append(("several deflate blocks (memLevel 1: 128 symbols per block)", (text * 8)[:40000], {"mem_level": 1}))
    # a skewed alphabet: code lengths up to 15 (longer than the tables' 10 / 8 index bits)
    p = 0.5 ** np.arange(1, 40)
    skew = rng.choice(39, size=60000, p=p / p.sum()).astype(np.uint8)
    out.append(("skewed literals: long codes", skew.tobytes(), {"strategy": zlib.Z_HUFFMAN_ONLY}))
    # matches at long distances: a 3 KiB phrase that comes back after 5, 9, 17 and 31 KiB of noise (beyond the ring; up to the window)
    phrase = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    far = phrase
    for gap in (5000, 9000, 17000, 27000):
        far += rng.integers(0, 256, gap, dtype=np.uint8).tobytes() + phrase
    out.append(("far matches", far[:65536], {"level": 9}))
    # distances of every small size with skewed lengths
    mix = bytearray()
    while len(mix) < 64000:
        d = int(rng.integers(1, 70))
        mix += rng.integers(0, 256, d, dtype=np.uint8).tobytes() * int(rng.integers(1, 9))
    out.append(("short periods", bytes(mix[:64000]), {"level": 6}))
    out.append(("max block", (recs * 2)[:65536], {"level": 4}))
    # many short matches with short codes (round 6's group decode: up to a few dozen symbols in one 64-bit buffer, more matches in a group
    # than the batch has room left for), and literals between long matches at every offset of the ring's 512-byte units
    out.append(("binary noise: dense short matches", rng.integers(0, 2, 65000, dtype=np.uint8).tobytes(), {"level": 9}))
    out.append(("four letters, level 1", rng.integers(0, 4, 65000, dtype=np.uint8).tobytes(), {"level": 1}))
    steps = bytearray()
    k = 0
    while len(steps) < 64000:
        steps += bytes([65 + k % 26]) * (200 + 13 * (k % 23)) + rng.integers(0, 256, k % 5, dtype=np.uint8).tobytes()
        k += 1
    out.append(("runs of 200-500 with a few literals between", bytes(steps[:64000]), {"level": 6}))
    return out


def test_inflate_every_kind_of_block_matches_zlib(ctx):
    rng = np.random.default_rng(7)
    items = corpus(rng)
    payloads = [deflate_raw(raw, **kw) for _, raw, kw in items]
    for p, (_, raw, _) in zip(payloads, items):
        assert zlib.decompress(p, wbits=-15) == raw
    # (the corpus really holds what it says: block types by their first header bits)
    assert payloads[10][0] & 6 == 0 and payloads[11][0] & 6 == 2 and payloads[3][0] & 6 == 4
    for trial in range(4):                                        # every alignment of input and output
        gaps = [int(rng.integers(0, 9)) + (trial if i == 0 else 0) for i in range(len(items))]
        out_gaps = [int(rng.integers(0, 40)) if trial else 0 for _ in items]
        status, outs = inflate_on_device(ctx, payloads, [raw for _, raw, _ in items], gaps, out_gaps)
        for (name, raw, _), st, got in zip(items, status, outs):
            assert st == 0, f"{name}: status {st:#x} (trial {trial})"
            assert got == raw, f"{name}: inflated bytes differ (trial {trial})"


def test_inflate_many_random_blocks(ctx):
    """a few thousand blocks of every size at once: the launch's dealing of blocks to waves, the ring at every phase"""
    rng = np.random.default_rng(11)
    raws, payloads = [], []
    base = rng.integers(0, 256, 200000, dtype=np.uint8)
    words = [rng.integers(65, 91, int(rng.integers(2, 12)), dtype=np.uint8).tobytes() for _ in range(300)]
    for i in range(3000):
        kind = i % 4
        n = int(rng.integers(0, 65537)) if i % 50 else 65536
        if kind == 0:
            o = int(rng.integers(0, 100000))
            raw = base[o:o + n].tobytes()
        elif kind == 1:
            raw = b" ".join(words[int(j)] for j in rng.integers(0, 300, n // 6 + 1))[:n]
        elif kind == 2:
            raw = bytes(rng.integers(0, 5, n // 9 + 1, dtype=np.uint8).repeat(9))[:n]
        else:
            raw = (base[:257].tobytes() * (n // 257 + 1))[:n]
        raws.append(raw)
        payloads.append(deflate_raw(raw, level=int(rng.integers(1, 10))))
    status, outs = inflate_on_device(ctx, payloads, raws, gaps=[int(g) for g in rng.integers(18, 44, len(raws))])
    assert (status == 0).all(), np.flatnonzero(status)[:10]
    for i, (raw, got) in enumerate(zip(raws, outs)):
        assert got == raw, f"block {i} ({len(raw)} bytes)"


def test_inflate_reports_damage_and_never_decides(ctx):
    """Bit flips, truncation, wrong sizes, wrong CRC: every damaged block has a non-zero status or inflates to exactly what zlib makes
    of the same bytes with the same CRC verdict; intact neighbours are untouched by it."""
    rng = np.random.default_rng(13)
    good = [rng.integers(0, 8, 20000, dtype=np.uint8).tobytes(), b"the quick brown fox " * 2000, bytes(30000)]
    payloads, raws, crcs, expect_ok = [], [], [], []
    for rep in range(120):
        raw = good[rep % 3]
        p = bytearray(deflate_raw(raw, level=1 + rep % 9))
        kind = rep % 6
        out_len = len(raw)
        crc = zlib.crc32(raw) & 0xFFFFFFFF
        if kind == 0:
            p[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8))       # one bit
        elif kind == 1:
            p = p[:int(rng.integers(1, len(p)))]                                   # cut short
        elif kind == 2:
            out_len += int(rng.integers(1, 100))                                   # ISIZE says more
        elif kind == 3:
            out_len -= int(rng.integers(1, 100))                                   # ISIZE says less
        elif kind == 4:
            crc ^= 0x10                                                            # wrong CRC
        # kind 5: intact
        payloads.append(bytes(p)); raws.append(bytes(out_len)); crcs.append(crc)
        # what zlib says about these bytes
        try:
            dz = zlib.decompressobj(wbits=-15)
            got = dz.decompress(bytes(p)) + dz.flush()
            ok = dz.eof and len(got) == out_len and (zlib.crc32(got) & 0xFFFFFFFF) == crc
        except zlib.error:
            got, ok = None, False
        expect_ok.append((ok, got))
    status, outs = inflate_on_device(ctx, payloads, raws, gaps=[3] * len(raws), crcs=crcs)
    for i, ((ok, got), st, out) in enumerate(zip(expect_ok, status, outs)):
        if st == 0:
            assert ok and out == got, f"block {i}: the device accepted what zlib does not (or other bytes)"
        if ok:
            assert st == 0 and out == got, f"block {i}: intact by zlib's verdict, status {st:#x}"
    assert sum(1 for st in status if st != 0) >= 60


def bam_stream(rng, n_records, n_ref=3):
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chr1\tLN:1000\n"
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_ref)
    for r in range(n_ref):
        nm = b"chr%d\0" % (r + 1)
        raw += struct.pack("<i", len(nm)) + nm + struct.pack("<i", 1 << 20)
    first = len(raw)
    recs = []
    for i in range(n_records):
        flag = int(rng.choice([99, 147, 83, 163, 1123, 4, 355, 2147, 65, 129, 73, 1, 97]))
        tid = int(rng.integers(-1, n_ref)); mtid = tid if rng.random() < 0.9 else int(rng.integers(-1, n_ref))
        tlen = int(rng.choice([0, 1, -1, 170, -170, 5000, 5001, -5000, -(1 << 31), (1 << 31) - 1, int(rng.integers(-6000, 6000))]))
        l_seq = int(rng.integers(0, 400))
        name = b"q%d\0" % i
        body = struct.pack("<iiBBHHHiiii", tid, i, len(name), 60, 4680, 1, flag, l_seq, mtid, i + 5, tlen) + name + struct.pack("<I", (l_seq << 4))
        body += rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8).tobytes() + rng.integers(0, 42, l_seq, dtype=np.uint8).tobytes()
        if i % 97 == 0:
            body += b"XZZ" + bytes(int(rng.integers(0, 70000)))      # a record longer than a BGZF block
        raw += struct.pack("<i", len(body)) + body
        recs.append(dict(flag=flag, refID=tid, next_refID=mtid, tlen=tlen))
    return raw, first, recs


def cut_blocks(raw, first, rng, mode):
    """block ends in the stream: mode 'records' = a block is flushed rather than a record split (htslib); 'anywhere'; 'tiny' = blocks of a few bytes"""
    ends = []
    if mode == "records":
        o, last = first, 0
        while o < len(raw):
            (bs,) = struct.unpack_from("<I", raw, o)
            nxt = o + 4 + bs
            if nxt - last > 0xff00 and o > last:
                ends.append(o); last = o
            while nxt - last > 0x10000:                           # a record longer than a block is split
                last += 0x10000; ends.append(last)
            o = nxt
        ends.append(len(raw))
    else:
        o = 0
        while o < len(raw):
            o = min(len(raw), o + int(rng.integers(1, 40) if mode == "tiny" and rng.random() < 0.5 else rng.integers(1000, 65536)))
            ends.append(o)
    return np.array(ends, dtype=np.uint64)


@pytest.mark.parametrize("mode", ["records", "anywhere", "tiny"])
def test_bam_walk_and_reduce(ctx, oracle, mode):
    rng = np.random.default_rng({"records": 1, "anywhere": 2, "tiny": 3}[mode])
    raw, first, recs = bam_stream(rng, 6000)
    ends = cut_blocks(raw, first, rng, mode)
    n = len(ends)
    # the entries a plain walk finds: for every block the first record that begins at or behind its beginning
    starts = []
    o = first
    while o < len(raw):
        starts.append(o)
        o += 4 + struct.unpack_from("<I", raw, o)[0]
    assert o == len(raw)
    starts = np.array(starts + [len(raw)], dtype=np.uint64)
    begins = np.concatenate([[0], ends[:-1]]).astype(np.uint64)
    want_entry = starts[np.searchsorted(starts, np.maximum(begins, first))]
    d = Dev(ctx)
    try:
        d_stream = d.put(np.frombuffer(raw, dtype=np.uint8))
        d_ends = d.put(ends.view(np.uint8))
        d_entry, d_exit, d_nrec = d.empty(8 * (n + 1)), d.empty(8 * (n + 1)), d.empty(4 * (n + 2))
        verified, n_records, rounds = ctx.bam_walk_dev(d_stream, len(raw), d_ends, n, first, d_entry, d_exit, d_nrec, max_rounds=100000, n_ref=3)
        assert verified and n_records == len(recs), (verified, n_records, rounds)
        assert (d.get(d_entry, n, np.uint64) == want_entry).all()
        # (blocks cut anywhere: the guesses are right, the rounds only confirm them; runs of blocks of a few bytes in which no record begins
        # are passed one block per round)
        assert rounds <= {"records": 3, "anywhere": 8, "tiny": 60}[mode], rounds
        max_frag = 5000
        d_out = d.put(np.zeros(4 + max_frag + 1, dtype=np.uint64).view(np.uint8))
        ctx.bam_walk_reduce_dev(d_stream, len(raw), d_ends, d_entry, n, max_frag, d_out)
        got = d.get(d_out, 4 + max_frag + 1, np.uint64)
        # a truncated stream, a record of impossible size: not verified
        for bad_len, bad_raw in ((len(raw) - 3, raw), (len(raw), raw[:first] + struct.pack("<I", 31) + raw[first + 4:])):
            d_bad = d.put(np.frombuffer(bad_raw, dtype=np.uint8))
            e2 = ends.copy(); e2[-1] = bad_len
            d_e2 = d.put(e2.view(np.uint8))
            v2, _, _ = ctx.bam_walk_dev(d_bad, bad_len, d_e2, n, first, d_entry, d_exit, d_nrec, max_rounds=100000)
            assert not v2
    finally:
        d.close()
    flag = np.array([r["flag"] for r in recs], dtype=np.uint16)
    tid = np.array([r["refID"] for r in recs], dtype=np.int32)
    mtid = np.array([r["next_refID"] for r in recs], dtype=np.int32)
    tlen = np.array([r["tlen"] for r in recs], dtype=np.int32)
    e_counters, e_hist, e_total = oracle.bam_flag_tlen(flag, tid, mtid, tlen, max_frag)
    assert (got[:3] == e_counters).all() and got[3] == e_total and (got[4:] == e_hist).all()
    spec = bam_spec.statistics([dict(flag=int(f)) for f in flag])
    assert tuple(int(x) for x in got[:3]) == tuple(spec)


def test_bam_file_reduce_matches_the_readers(ctx, oracle, tmp_path, monkeypatch):
    """sk_bam_file_reduce on files written three ways (blocks cut anywhere — tests' writer —, a file of one block, many small chunks of
    the pipeline) equals the specification's reader; files it must not handle are left to the caller."""
    rng = np.random.default_rng(5)
    refs = [("chr1", 100000), ("chr2", 50000)]
    records = []
    for i in range(30000):
        flag = int(rng.choice([99, 147, 83, 163, 1123, 4, 355, 2147, 65, 129]))
        tl = int(rng.integers(-6000, 6000))
        records.append(dict(tid=int(rng.integers(0, 2)), pos=i, flag=flag, mtid=int(rng.integers(0, 2)), mpos=i + 3, tlen=tl, name="r%d" % i,
                            seq_len=int(rng.integers(1, 200))))
    path = str(tmp_path / "a.bam")
    write_bam(path, refs, records)
    _, recs = bam_spec.read_bam(path)
    e_stats = bam_spec.statistics(recs)
    e_hist = bam_spec.fragment_lengths(recs, 5000)
    for chunk_log2 in (None, "12", "16", "novmm"):                # (the last: the inflated stream's room from hipMalloc, not a mapped range)
        if chunk_log2 == "novmm":
            monkeypatch.setenv("SK_BAMFILE_NO_VMM", "1")
        elif chunk_log2:
            monkeypatch.setenv("SK_BAMFILE_CHUNK_LOG2", chunk_log2)
        handled, counters, hist, total, info = ctx.bam_file_reduce(path, 5000)
        assert handled, info
        assert tuple(int(x) for x in counters) == tuple(e_stats)
        assert [int(x) for x in hist] == e_hist[0] and total == e_hist[1]
        assert info[3] == len(recs) and info[4] == 0
    monkeypatch.delenv("SK_BAMFILE_CHUNK_LOG2")
    monkeypatch.delenv("SK_BAMFILE_NO_VMM")
    # not handled: a truncated file, a file that is not BGZF, a missing file
    data = open(path, "rb").read()
    for name, blob in (("cut.bam", data[:len(data) // 2]), ("plain.bam", b"BAM\1" + bytes(100)), ("gz.bam", zlib.compress(b"BAM\1" + bytes(1000)))):
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        handled, counters, _, _, _ = ctx.bam_file_reduce(p, 5000)
        assert not handled and not counters.any()
    handled, _, _, _, _ = ctx.bam_file_reduce(str(tmp_path / "nope.bam"), 5000)
    assert not handled


def test_bam_file_reduce_keeps_and_regrows_its_buffers(ctx, oracle, tmp_path):
    """The call's buffers stay with the ctx (the compressed file's, the mapped range for the inflated stream): a small file, then one
    whose stream needs a bigger range than the first call reserved, then the small one again — every call equals the oracle."""
    from seqkit_amd import synth
    files = []
    for name, n_rec, unit, kind in (("small.bam", 20_000, 20_000, "sorted"), ("big.bam", 320_000, 80_000, "random"), ("small2.bam", 30_000, 30_000, "random")):
        path = str(tmp_path / name)
        n, flag, tid, mtid, tlen, reps = synth.write_bam_file(path, n_rec, seed=11, kind=kind, unit_records=unit)
        files.append((path, n, oracle.bam_flag_tlen(flag, tid, mtid, tlen, 5000), reps))
    assert os.path.getsize(files[1][0]) * 6 > 256 << 20            # (more than the first call's reservation)
    for path, n, (e_counters, e_hist, e_total), reps in files + files[::-1]:
        handled, counters, hist, total, info = ctx.bam_file_reduce(path, 5000)
        assert handled and info[3] == n and info[4] == 0
        assert (counters == e_counters * np.uint64(reps)).all() and (hist == e_hist * np.uint64(reps)).all() and total == e_total * reps


def test_inflate_across_4_gib_of_output(ctx):
    """A block whose output straddles 2^32 in the inflated stream (the 66 000th block of a 3.6 GB BAM): the kernel's address
    arithmetic is modular and must not notice."""
    rng = np.random.default_rng(17)
    raws = [rng.integers(0, 7, 65536, dtype=np.uint8).tobytes(), (b"ACGTTGCA" * 9000)[:65000], rng.integers(0, 256, 60000, dtype=np.uint8).tobytes()]
    pays = [deflate_raw(r, level=6) for r in raws]
    base = (1 << 32) - 100_000
    n = len(raws)
    blocks = np.zeros(n, dtype=ctx.BGZF_BLOCK_DTYPE)
    comp = bytearray()
    out_off = base + 7
    for i, (p, r) in enumerate(zip(pays, raws)):
        blocks[i] = (len(comp), len(p), len(r), out_off, zlib.crc32(r) & 0xFFFFFFFF, 0)
        comp += p + bytes(5)
        out_off += len(r)
    assert int(blocks[1]["out_off"]) < (1 << 32) < int(blocks[1]["out_off"]) + len(raws[1])
    d = Dev(ctx)
    try:
        d_comp = d.put(np.frombuffer(bytes(comp) + bytes(16), dtype=np.uint8))
        d_blocks = d.put(blocks.view(np.uint8))
        d_out = d.empty(out_off + 64)
        d_status = d.put(np.full(n, 0xFFFFFFFF, dtype=np.uint32).view(np.uint8))
        ctx.bgzf_inflate_dev(d_comp, d_blocks, n, d_out, d_status, True)
        ctx.sync()
        assert (d.get(d_status, n, np.uint32) == 0).all()
        got = d.get(d_out + base, out_off - base).tobytes()
    finally:
        d.close()
    assert got[7:] == b"".join(raws)
