#!/usr/bin/env python3
"""End-to-end (file -> stdout/files) wall time of the C++ hosts vs the oracle CLI on the GPU box.  Host/PCIe/zlib bound:
reported separately from the kernel numbers, never as bench.py's value."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from seqkit_amd import build, synth  # noqa: E402

orc.build()
build.build_all()
FASTA = os.environ.get("E2E_FASTA") or os.path.join(build.BINDIR, "fasta")     # E2E_FASTA: another build of the host, for A/B on one box
n_block, reps = 100_000, int(sys.argv[1]) if len(sys.argv) > 1 else 20
d = tempfile.mkdtemp(prefix="sk_e2e_")
seq, qual = synth.make_reads(n_block, 150, seed=1)
table = synth.make_sheet(96, 8, dual=True, seed=4)
bc, _ = synth.observe_barcodes(table, n_block, seed=4, halves=2)
headers = [f"@SIM:1:{i} 1:N:0".encode() + b" BC:" + bc[i].tobytes() for i in range(n_block)]
block = synth.fastq_text(seq, qual, headers=headers)
fq = os.path.join(d, "in.fq")
with open(fq, "wb") as f:
    for _ in range(reps):
        f.write(block)
sheet = os.path.join(d, "sheet.tsv")
with open(sheet, "wb") as f:
    for i in range(96):
        f.write(f"S{i:02d}\t".encode() + table[i].tobytes() + b"\n")
n = n_block * reps
print(f"{n} reads, {os.path.getsize(fq) / 1e6:.0f} MB FASTQ")


import resource  # noqa: E402


def t(cmd, cwd):
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=None if os.environ.get("E2E_STDERR") else subprocess.DEVNULL)
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    print(f"    cpu user {r1.ru_utime - r0.ru_utime:.1f} s sys {r1.ru_stime - r0.ru_stime:.1f} s", flush=True)
    return dt, r.returncode


ONLY = os.environ.get("E2E_ONLY", "")
for name, args in (("trim by quality", ["trim", "by", "quality", fq, "20"]), ("mask by quality", ["mask", "by", "quality", fq, "20"]),
                   ("demultiplex (96 samples, gz out)", ["demultiplex", sheet, fq]),
                   ("demultiplex --dry-run (census)", ["demultiplex", f"--dry-run={n}", sheet, fq]), ("statistics (census)", ["statistics", fq])):
    if ONLY and ONLY not in name:
        continue
    for label, binary in (("hip", FASTA),) if os.environ.get("E2E_NO_ORACLE") else (("hip", FASTA), ("oracle", orc.FASTA_BIN)):
        w = os.path.join(d, label + name.split()[0])
        os.makedirs(w, exist_ok=True)
        dt, rc = t([binary] + args, w)
        print(f"{name:34s} {label:7s} {dt:7.2f} s  {n / dt / 1e6:6.2f} M reads/s  rc={rc}", flush=True)


# compressed inputs: one gzip stream (inflated by one zlib stream, as `gunzip -c` would) against BGZF (what this build's
# GzWriter and bgzip write: inflated block-parallel)
if ONLY and ONLY in "gz inputs":
    import gzip
    import struct
    import zlib

    def bgzf(data):
        out = []
        for o in list(range(0, len(data), 0xff00)) + [None]:
            piece = b"" if o is None else data[o:o + 0xff00]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            comp = c.compress(piece) + c.flush()
            out.append(struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(piece), len(piece)))
        return b"".join(out)

    gz1 = os.path.join(d, "plain.fq.gz")
    gz2 = os.path.join(d, "bgzf.fq.gz")
    one = gzip.compress(block, 1)
    with open(gz1, "wb") as f:                       # `reps` members: any gzip reader goes through them one after the other
        for _ in range(reps):
            f.write(one)
    blk = bgzf(block)[:-28]
    with open(gz2, "wb") as f:
        for _ in range(reps):
            f.write(blk)
        f.write(bgzf(b""))
    for label, path in (("gzip members", gz1), ("BGZF", gz2)):
        w = os.path.join(d, "gzin")
        os.makedirs(w, exist_ok=True)
        dt, rc = t([FASTA, "trim", "by", "quality", path, "20"], w)
        print(f"{'trim by quality, input ' + label:40s} {dt:7.2f} s  {n / dt / 1e6:6.2f} M reads/s  rc={rc}", flush=True)

# the reference README's pipeline (BASELINE configs[3]): both mates through `add barcode`, joined by `demultiplex`
if not ONLY or ONLY in "pipeline":
    plain = [f"@SIM:1:{i} 1:N:0".encode() for i in range(n_block)]
    r1 = os.path.join(d, "R1.fq")
    r2 = os.path.join(d, "R2.fq")
    i1 = os.path.join(d, "I1.fq")
    seq2, qual2 = synth.make_reads(n_block, 150, seed=2)
    b1 = synth.fastq_text(seq, qual, headers=plain)
    b2 = synth.fastq_text(seq2, qual2, headers=plain)
    bi = b"".join(plain[i] + b"\n" + bc[i].tobytes() + b"\n+\n" + b"I" * bc.shape[1] + b"\n" for i in range(n_block))
    for path, blk in ((r1, b1), (r2, b2), (i1, bi)):
        with open(path, "wb") as f:
            for _ in range(reps):
                f.write(blk)
    for label, binary in (("hip", FASTA),) if os.environ.get("E2E_NO_ORACLE") else (("hip", FASTA), ("oracle", orc.FASTA_BIN)):
        w = os.path.join(d, label + "pipe")
        os.makedirs(w, exist_ok=True)
        dt, rc = t(["bash", "-c", f"{binary} demultiplex {sheet} <({binary} add barcode {r1} {i1}) <({binary} add barcode {r2} {i1})"], w)
        print(f"{'add barcode x2 | demultiplex (pairs)':34s} {label:7s} {dt:7.2f} s  {n / dt / 1e6:6.2f} M pairs/s  rc={rc}", flush=True)

# f4: sam to fastq on a BAM of paired 150 bp reads
if not ONLY or ONLY in "sam to fastq":
    import struct
    import zlib
    SAM = os.path.join(build.BINDIR, "sam")
    n_bam = 100_000 * min(reps, 10)
    rng = np.random.default_rng(3)
    bam = os.path.join(d, "in.bam")
    codes = np.array([1, 2, 4, 8], dtype=np.uint8)

    def bgzf(data):
        c = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = c.compress(data) + c.flush()
        return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))

    with open(bam, "wb") as f:
        text = b"@HD\tVN:1.6\n"
        f.write(bgzf(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1 << 28)))
        buf = bytearray()
        for i in range(n_bam // 2):
            for mate in (0, 1):
                name = b"read%d\0" % i
                nib = codes[rng.integers(0, 4, size=150)]
                packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
                q = rng.integers(2, 41, size=150, dtype=np.uint8).tobytes()
                flag = 1 | (64 if mate == 0 else 128) | (16 if (i + mate) % 2 else 0)
                body = struct.pack("<iiBBHHHiiii", 0, i, len(name), 60, 4680, 1, flag, 150, 0, i, 0) + name + struct.pack("<I", 150 << 4) + packed + q
                buf += struct.pack("<i", len(body)) + body
                if len(buf) > 60000:
                    f.write(bgzf(bytes(buf[:60000])))
                    del buf[:60000]
        while buf:
            f.write(bgzf(bytes(buf[:60000])))
            del buf[:60000]
        f.write(bgzf(b""))
    print(f"{n_bam} BAM records, {os.path.getsize(bam) / 1e6:.0f} MB")
    for label, binary in (("hip", SAM),) if os.environ.get("E2E_NO_ORACLE") else (("hip", SAM), ("oracle", orc.SAM_BIN)):
        w = os.path.join(d, label + "sam")
        os.makedirs(w, exist_ok=True)
        dt, rc = t([binary, "to", "fastq", bam, "out"], w)
        print(f"{'sam to fastq (gz out)':34s} {label:7s} {dt:7.2f} s  {n_bam / dt / 1e6:6.2f} M reads/s  rc={rc}", flush=True)
        dt, rc = t([binary, "to", "interleaved", "fastq", bam], w)
        print(f"{'sam to interleaved fastq':34s} {label:7s} {dt:7.2f} s  {n_bam / dt / 1e6:6.2f} M reads/s  rc={rc}", flush=True)
        for cmd in (["statistics", bam], ["fragment", "lengths", bam], ["fragments", bam]):
            dt, rc = t([binary] + cmd, w)
            print(f"{'sam ' + ' '.join(cmd[:-1]):34s} {label:7s} {dt:7.2f} s  {n_bam / dt / 1e6:6.2f} M records/s  rc={rc}", flush=True)
