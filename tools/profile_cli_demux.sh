#!/bin/bash
# Kernel trace of the `fasta demultiplex` host on cfg 3 text (16 single-index 8 bp barcodes in the header's BC: field): which
# kernel the plain command runs.  usage: bash tools/profile_cli_demux.sh <tag> [reads]   -> gpurun_out/<tag>/cli_demux_*
set -u
TAG=${1:-cli}; N=${2:-2000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
W=/tmp/skcli_$TAG
rm -rf $W; mkdir -p $W/run $OUT
cd $W && export TMPDIR=/tmp
python3 - "$R" "$N" <<'PY'
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from seqkit_amd import synth
n = int(sys.argv[2]); blk = 100_000
table = synth.make_sheet(16, 8, dual=False, seed=3)
with open("sheet.tsv", "wb") as f:
    for i in range(16):
        f.write(f"S{i:02d}\t".encode() + table[i].tobytes() + b"\n")
seq, qual = synth.make_reads(blk, 150, seed=3)
with open("reads.fq", "wb") as f:
    for b0 in range(0, n, blk):
        bc, _ = synth.observe_barcodes(table, blk, seed=3 + b0)
        headers = [f"@SIM:3:{b0 + i} 1:N:0".encode() + b" BC:" + bc[i].tobytes() for i in range(blk)]
        f.write(synth.fastq_text(seq, qual, headers=headers))
print(n, "reads written")
PY
cd $W/run
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -- $R/seqkit_amd/bin/fasta demultiplex ../sheet.tsv ../reads.fq > $W/trace.log 2> $W/trace.err
tail -2 $W/trace.err
python3 $R/tools/summarize_prof.py $W/trace $OUT/cli_demux
cat $OUT/cli_demux_kernel_stats.csv | cut -c1-200
