"""A third, independent decode of BAM for the parity tests — written from the SAM/BAM specification (SAMv1 §4.1 BGZF,
§4.2 BAM), NOT from this repo's readers (seqkit_amd/csrc/sam_main.cpp, oracle/sam_oracle_main.c) and not from the
tests' own writer: a shared misreading of the format in those three would pass their mutual comparisons; it would not
pass this one.  Plain Python: struct + zlib.

The reference reads BAM through rust-htslib 0.31 (src/common.rs:121-157); what it then does with a record is restated
here straight from the cited reference lines, in Python, for the two reductions of the hot path.
"""
import struct
import zlib


def bgzf_blocks(data: bytes):
    """SAMv1 §4.1: a BGZF file is a series of gzip members (RFC 1952) whose extra field holds the subfield SI1='B' SI2='C'
    SLEN=2 with BSIZE = total block size - 1.  Yields the inflated payload of every block, CRC32 and ISIZE checked."""
    at = 0
    while at < len(data):
        id1, id2, cm, flg, _mtime, _xfl, _os = struct.unpack_from("<BBBBIBB", data, at)
        assert (id1, id2, cm) == (31, 139, 8) and flg & 4, "not a BGZF block"
        (xlen,) = struct.unpack_from("<H", data, at + 10)
        extra = data[at + 12:at + 12 + xlen]
        bsize = None
        k = 0
        while k + 4 <= len(extra):                      # RFC 1952 extra subfields: SI1 SI2 SLEN(2) data
            si1, si2, slen = struct.unpack_from("<BBH", extra, k)
            if (si1, si2, slen) == (66, 67, 2):
                (bsize,) = struct.unpack_from("<H", extra, k + 4)
            k += 4 + slen
        assert bsize is not None, "BGZF block without a BC subfield"
        total = bsize + 1
        cdata = data[at + 12 + xlen:at + total - 8]
        crc, isize = struct.unpack_from("<II", data, at + total - 8)
        raw = zlib.decompress(cdata, wbits=-15)         # raw DEFLATE
        assert len(raw) == isize and (zlib.crc32(raw) & 0xFFFFFFFF) == crc
        yield raw
        at += total


def read_bam(path):
    """SAMv1 §4.2.  Returns (references [(name, length)], records) where a record is a dict of the fixed-length fields."""
    with open(path, "rb") as f:
        raw = b"".join(bgzf_blocks(f.read()))
    assert raw[:4] == b"BAM\x01"
    (l_text,) = struct.unpack_from("<i", raw, 4)
    at = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", raw, at)
    at += 4
    refs = []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", raw, at)
        name = raw[at + 4:at + 4 + l_name - 1].decode()
        (l_ref,) = struct.unpack_from("<i", raw, at + 4 + l_name)
        refs.append((name, l_ref))
        at += 8 + l_name
    recs = []
    while at < len(raw):
        (block_size,) = struct.unpack_from("<i", raw, at)
        # refID, pos, l_read_name, mapq, bin, n_cigar_op, flag, l_seq, next_refID, next_pos, tlen
        ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, next_ref_id, next_pos, tlen = struct.unpack_from("<iiBBHHHIiii", raw, at + 4)
        name = raw[at + 36:at + 36 + l_read_name - 1]
        recs.append(dict(refID=ref_id, pos=pos, mapq=mapq, flag=flag, l_seq=l_seq, next_refID=next_ref_id, next_pos=next_pos, tlen=tlen,
                         n_cigar=n_cigar, name=name))
        at += 4 + block_size
    return refs, recs


# SAMv1 §1.4 FLAG bits
PAIRED, UNMAPPED, MATE_UNMAPPED, FIRST, SECONDARY, DUPLICATE, SUPPLEMENTARY = 0x1, 0x4, 0x8, 0x40, 0x100, 0x400, 0x800


def statistics(recs):
    """src/sam_statistics.rs:63-69: secondary / supplementary records are skipped; total; unmapped skipped; aligned; duplicates."""
    total = aligned = dup = 0
    for r in recs:
        f = r["flag"]
        if f & SECONDARY or f & SUPPLEMENTARY:
            continue
        total += 1
        if f & UNMAPPED:
            continue
        aligned += 1
        if f & DUPLICATE:
            dup += 1
    return total, aligned, dup


def fragment_lengths(recs, max_frag=5000, reads=None):
    """src/sam_fragment_lengths.rs:29-43: paired, first mate, both mates mapped, not duplicate / secondary / supplementary, same
    reference; |tlen| <= max; histogram; stop after `reads` kept records."""
    hist = [0] * (max_frag + 1)
    total = 0
    for r in recs:
        f = r["flag"]
        if not f & PAIRED or not f & FIRST:
            continue
        if f & UNMAPPED or f & MATE_UNMAPPED or f & DUPLICATE or f & SECONDARY or f & SUPPLEMENTARY:
            continue
        if r["refID"] != r["next_refID"]:
            continue
        frag = abs(r["tlen"])
        if frag > max_frag:
            continue
        total += 1
        hist[frag] += 1
        if reads is not None and total >= reads:
            break
    return hist, total


def pct(a, b):
    """Rust `{:.1}` of a / b * 100 (src/sam_statistics.rs:110-111); 0/0 prints NaN."""
    if b == 0:
        return "NaN"
    return f"{a / b * 100.0:.1f}"


def statistics_text(recs) -> bytes:
    total, aligned, dup = statistics(recs)
    return (f"Total reads: {total}\nAligned reads: {aligned} ({pct(aligned, total)}% of all reads)\n"
            f"Duplicate reads: {dup} ({pct(dup, aligned)}% of aligned reads)\n").encode()


def fragment_lengths_text(recs, max_frag=5000, reads=None) -> bytes:
    hist, _ = fragment_lengths(recs, max_frag, reads)
    return "".join(f"{size}\t{hist[size]}\n" for size in range(1, max_frag + 1)).encode()
