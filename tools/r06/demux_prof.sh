#!/bin/bash
# `fasta demultiplex` of 8 M reads into 96 .gz files with SEQKIT_PROF=1 (where the main and the reader thread's time went), and the same input through
# `trim by quality` and `demultiplex --dry-run` for scale.   usage (GPU box): bash tools/r06/demux_prof.sh <tag> [blocks of 100 k reads]
set -u
TAG=$1; REPS=${2:-80}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
{
  ENVS=("" "" "SEQKIT_SLOW_EXIT=1" "SEQKIT_MALLOC_DEFAULT=1" "SEQKIT_GPU_DEFLATE=0")
  for ENV in "${ENVS[@]}"; do
    echo "== demultiplex $ENV"
    env $ENV SEQKIT_PROF=1 E2E_STDERR=1 E2E_NO_ORACLE=1 E2E_ONLY="demultiplex (96" timeout -k 10 600 python3 tools/cli_e2e.py $REPS 2>&1 | grep -v "amdgpu.ids\|clusters carried\|Reading sample\|Starting demul" | tail -10
  done
  echo "== trim by quality / dry run"
  SEQKIT_PROF=1 E2E_STDERR=1 E2E_NO_ORACLE=1 E2E_ONLY="trim by" timeout -k 10 600 python3 tools/cli_e2e.py $REPS 2>&1 | grep -v amdgpu.ids | tail -4
  SEQKIT_PROF=1 E2E_STDERR=1 E2E_NO_ORACLE=1 E2E_ONLY="dry-run" timeout -k 10 600 python3 tools/cli_e2e.py $REPS 2>&1 | grep -v "amdgpu.ids\|^- " | tail -6
} 2>&1 | tee $OUT/demux_prof.txt
