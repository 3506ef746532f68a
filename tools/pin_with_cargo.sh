#!/bin/bash
# Pin the oracle to the REAL reference: on a host that has Rust, build annalam/seqkit v0.8.0 and diff its `fasta` / `sam` binaries
# against oracle/fasta_oracle and oracle/sam_oracle on the corpus of tools/pin_cases.py (SURVEY.md Appendix A's vectors, cfg 1-3
# inputs, the text-layer cases of tests/test_text_model.py, small BAMs).
#
# THIS SCRIPT HAS NEVER BEEN RUN: the build image has no cargo / rustc and no network (parity unpinned).  It is the
# one command that would lift that: a clean exit means every case's stdout, stderr, exit code and decompressed output files are
# byte-identical between the reference and the oracle.
#
# usage: tools/pin_with_cargo.sh <path to a checkout of annalam/seqkit>
set -euo pipefail
REF=${1:?usage: tools/pin_with_cargo.sh <seqkit checkout>}
command -v cargo > /dev/null || { echo "cargo not found: this host cannot build the reference" >&2; exit 2; }
( cd "$REF" && cargo build --release )
HERE=$(cd "$(dirname "$0")/.." && pwd)
make -C "$HERE/oracle" > /dev/null
python3 "$HERE/tools/pin_cases.py" --ref-fasta "$REF/target/release/fasta" --ref-sam "$REF/target/release/sam"
