#!/bin/bash
# `fasta demultiplex` of 8 M reads into 96 .gz files: CPU deflate (the pool of threads, SEQKIT_GPU_DEFLATE=0) against the device's (the default);
# the decompressed outputs of both are compared.   usage (GPU box): bash tools/r06/deflate_e2e.sh <out dir> [blocks of 100 k reads]
set -u
OUT=$1; REPS=${2:-80}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
{
  echo "== CPU deflate (SEQKIT_GPU_DEFLATE=0)"
  SEQKIT_GPU_DEFLATE=0 E2E_NO_ORACLE=1 E2E_ONLY="demultiplex (96" timeout -k 10 600 python3 tools/cli_e2e.py $REPS 2>&1 | tail -4
  echo "== device deflate (the default)"
  SEQKIT_GPU_DEFLATE=1 E2E_NO_ORACLE=1 E2E_ONLY="demultiplex (96" timeout -k 10 600 python3 tools/cli_e2e.py $REPS 2>&1 | tail -4
} | tee $OUT/deflate_e2e.txt
