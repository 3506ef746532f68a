"""Helpers for the command-line parity tests: run a binary, build FASTQ / sample-sheet / BAM inputs."""
import gzip
import os
import struct
import subprocess
import zlib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FASTA = os.path.join(REPO, "seqkit_amd", "bin", "fasta")
SAM = os.path.join(REPO, "seqkit_amd", "bin", "sam")


def run(binary, args, cwd=None, stdin=None, env=None):
    e = dict(os.environ, **env) if env else None
    r = subprocess.run([binary] + list(args), cwd=cwd, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=e)
    return r.returncode, r.stdout, r.stderr


def gunzip_dir(d):
    """{file name: decompressed bytes} of every *.gz in a directory (parity is on the decompressed streams)."""
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".gz"):
            out[f] = gzip.open(os.path.join(d, f), "rb").read()
    return out


def bgzf_block(data: bytes) -> bytes:
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    bsize = len(comp) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    return hdr + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def write_bam(path, refs, records, truncate=None):
    """refs: [(name, length)], records: dicts with tid,pos,flag,mtid,mpos,tlen,name,cigar[(op,len)],seq_len; optional
    `codes` (one 4-bit BAM base code per base) and `qual` (raw phred per base) give the record real bases; `name` may be
    bytes."""
    raw = b"BAM\1"
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    raw += struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for name, ln in refs:
        nb = name.encode() + b"\0"
        raw += struct.pack("<i", len(nb)) + nb + struct.pack("<i", ln)
    for r in records:
        name = r.get("name", "q")
        name = (name if isinstance(name, bytes) else name.encode()) + b"\0"
        codes = r.get("codes")
        l_seq = len(codes) if codes is not None else r.get("seq_len", 10)
        cigar = r.get("cigar", [(0, l_seq)])
        body = struct.pack("<iiBBHHHiiii", r["tid"], r["pos"], len(name), r.get("mapq", 60), 4680, len(cigar), r["flag"], l_seq,
                           r["mtid"], r["mpos"], r["tlen"])
        body += name + b"".join(struct.pack("<I", (ln << 4) | op) for op, ln in cigar)
        if codes is None:
            body += bytes((l_seq + 1) // 2) + bytes([30] * l_seq)
        else:
            cs = list(codes) + [0]
            body += bytes((cs[2 * k] << 4) | cs[2 * k + 1] for k in range((l_seq + 1) // 2)) + bytes(r.get("qual", [30] * l_seq))
        raw += struct.pack("<i", len(body)) + body
    if truncate is not None:
        raw = raw[:truncate]
    with open(path, "wb") as f:
        for i in range(0, len(raw), 60000):
            f.write(bgzf_block(raw[i:i + 60000]))
        f.write(bgzf_block(b""))
