#!/usr/bin/env python3
"""File -> stdout wall time of the `sam` host's reductions on a BAM of many records (the same 100 k paired records over
and over: the content does not matter to the readers), next to the oracle command line.  usage: bam_e2e.py [million records]"""
import os
import struct
import subprocess
import sys
import tempfile
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from seqkit_amd import build  # noqa: E402

orc.build()
build.build_all()
SAM = os.path.join(build.BINDIR, "sam")
millions = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(3)
codes = np.array([1, 2, 4, 8], dtype=np.uint8)
d = tempfile.mkdtemp(prefix="sk_bam_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
bam = os.path.join(d, "in.bam")


def bgzf(data, level=1):
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(data) + c.flush()
    return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


unit = bytearray()
ends = []                                # where records end inside the unit
for i in range(50_000):
    tl = int(rng.lognormal(np.log(170), 0.35))
    for mate in (0, 1):
        name = b"read%d\0" % i
        nib = codes[rng.integers(0, 4, size=150)]
        packed = ((nib[0::2] << 4) | nib[1::2]).astype(np.uint8).tobytes()
        q = rng.integers(2, 41, size=150, dtype=np.uint8).tobytes()
        flag = 1 | 2 | (64 | 32 if mate == 0 else 128 | 16)
        body = struct.pack("<iiBBHHHiiii", 0, i, len(name), 60, 4680, 1, flag, 150, 0, i + (tl if mate == 0 else -tl), tl if mate == 0 else -tl) + name + struct.pack("<I", 150 << 4) + packed + q
        unit += struct.pack("<i", len(body)) + body
        ends.append(len(unit))
unit = bytes(unit)
if os.environ.get("BAM_STRADDLE"):
    blocks = [bgzf(unit[o:o + 60000]) for o in range(0, len(unit), 60000)]   # blocks cut anywhere: records (and their heads) straddle them
else:
    # what htslib writes (bgzf_flush_try): a block is flushed rather than a record split, so every block begins with a record
    blocks, lo, prev = [], 0, 0
    for e in ends:
        if e - lo > 0xff00:
            blocks.append(bgzf(unit[lo:prev]))
            lo = prev
        prev = e
    blocks.append(bgzf(unit[lo:]))
t0 = time.perf_counter()
with open(bam, "wb") as f:
    text = b"@HD\tVN:1.6\n"
    f.write(bgzf(b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1 << 28)))
    body = b"".join(blocks)
    for _ in range(millions * 10):
        f.write(body)
    f.write(bgzf(b""))
n = millions * 1_000_000
print(f"{n} BAM records, {os.path.getsize(bam) / 1e6:.0f} MB, written in {time.perf_counter() - t0:.1f} s", flush=True)
for label, binary in (("hip", SAM),) if os.environ.get("E2E_NO_ORACLE") else (("hip", SAM), ("oracle", orc.SAM_BIN)):
    for cmd in (["statistics", bam], ["fragment", "lengths", bam], ["fragments", bam], ["to", "interleaved", "fastq", bam]):
        if label == "oracle" and cmd[0] != "statistics":
            continue
        t0 = time.perf_counter()
        r = subprocess.run([binary] + cmd, stdout=subprocess.DEVNULL, stderr=None if os.environ.get("E2E_STDERR") else subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        print(f"{'sam ' + ' '.join(cmd[:-1]):26s} {label:7s} {dt:7.2f} s  {n / dt / 1e6:7.2f} M records/s  rc={r.returncode}", flush=True)
if os.environ.get("BAM_KEEP"):
    import shutil
    shutil.move(bam, os.environ["BAM_KEEP"])         # for tools/bam_scale.sh
else:
    os.remove(bam)
os.rmdir(d)
