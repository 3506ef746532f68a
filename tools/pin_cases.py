#!/usr/bin/env python3
"""The corpus of tools/pin_with_cargo.sh: run the reference's binaries and the oracle's on the same inputs, compare everything
observable (stdout, stderr, exit code, decompressed *.gz outputs).  Never run here (no Rust toolchain); see the shell script.
Known, documented differences are normalised: a Rust panic's message text (exit 101: only the code is compared), and the order
of equal counts in the dry run's table (a HashMap's iteration order)."""
import argparse
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from hypothesis import HealthCheck, given, settings  # noqa: E402

from oracle import oracle as orc  # noqa: E402
from seqkit_amd import synth  # noqa: E402
from tests import cli_util as cu  # noqa: E402
from tests.test_text_model import cases  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ref-fasta", required=True)
ap.add_argument("--ref-sam", required=True)
args = ap.parse_args()
failures = 0


def run_both(ref_bin, orc_bin, argv, files, label):
    global failures
    outs = []
    for binary in (ref_bin, orc_bin):
        d = tempfile.mkdtemp(prefix="sk_pin_")
        try:
            for name, data in files.items():
                with open(os.path.join(d, name), "wb") as f:
                    f.write(data)
            r = subprocess.run([binary] + argv, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            gz = {f: gzip.open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d)) if f.endswith(".gz")}
        finally:
            shutil.rmtree(d, ignore_errors=True)
        outs.append((r.returncode % 256, r.stdout, r.stderr, gz))
    a, b = outs
    same = a[0] == b[0] and (a[0] == 101 or (sorted(a[1].splitlines()) == sorted(b[1].splitlines()) and a[2] == b[2] and a[3] == b[3]))
    if not same:
        failures += 1
        print(f"DIFFERENT: {label}: {' '.join(argv)}\n  reference rc={a[0]} stderr={a[2][-300:]!r}\n  oracle    rc={b[0]} stderr={b[2][-300:]!r}")


# cfg 1-3 shaped inputs
seq, qual = synth.make_reads(2000, 150, seed=1)
qual = synth.add_forced_classes(qual, seed=2)
fq = synth.fastq_text(seq, qual, prefix="SIM:1")
for m in ("0", "2", "20", "30", "41", "255", "256", "x"):
    run_both(args.ref_fasta, orc.FASTA_BIN, ["trim", "by", "quality", "r.fq", m], {"r.fq": fq}, "trim")
    run_both(args.ref_fasta, orc.FASTA_BIN, ["mask", "by", "quality", "r.fq", m], {"r.fq": fq}, "mask")
table = synth.make_sheet(16, 8, seed=3)
bc, _ = synth.observe_barcodes(table, 2000, seed=3)
headers = [f"@SIM:3:{i} 1:N:0 BC:".encode() + bc[i].tobytes() for i in range(2000)]
sheet = b"".join(f"S{i}\t".encode() + table[i].tobytes() + b"\n" for i in range(16))
run_both(args.ref_fasta, orc.FASTA_BIN, ["demultiplex", "sheet.tsv", "r.fq"], {"sheet.tsv": sheet, "r.fq": synth.fastq_text(seq, qual, headers=headers)}, "demultiplex cfg3")
run_both(args.ref_fasta, orc.FASTA_BIN, ["demultiplex", "--dry-run=1500", "sheet.tsv", "r.fq"], {"sheet.tsv": sheet, "r.fq": synth.fastq_text(seq, qual, headers=headers)}, "dry run")


# the text layer
@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(cases())
def text_cases(case):
    files, argv, *_ = case
    run_both(args.ref_fasta, orc.FASTA_BIN, ["demultiplex"] + argv, files, "text layer")


text_cases()
# BAM
rng = np.random.default_rng(5)
recs = [dict(tid=int(rng.integers(0, 2)), pos=i, flag=int(rng.choice([99, 147, 83, 163, 1123, 4, 355, 2147, 65, 129])), mtid=int(rng.integers(0, 2)), mpos=i + 3,
             tlen=int(rng.integers(-6000, 6000)), name=f"r{i}", seq_len=int(rng.integers(1, 200))) for i in range(20000)]
d = tempfile.mkdtemp(prefix="sk_pin_bam_")
cu.write_bam(os.path.join(d, "a.bam"), [("chr1", 100000), ("chr2", 50000)], recs)
bam = open(os.path.join(d, "a.bam"), "rb").read()
shutil.rmtree(d)
for argv in (["statistics", "a.bam"], ["fragment", "lengths", "a.bam"], ["fragment", "lengths", "--max-frag-size=300", "--reads=1000", "a.bam"], ["fragments", "a.bam"]):
    run_both(args.ref_sam, orc.SAM_BIN, argv, {"a.bam": bam}, "sam")
print(f"{failures} differences")
sys.exit(1 if failures else 0)
