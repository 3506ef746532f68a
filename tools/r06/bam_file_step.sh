#!/bin/bash
# sk_bam_file_reduce on the 3.6 GB file under the pipeline's knobs + the CLI.  usage (GPU box): bash tools/r06/bam_file_step.sh <tag> [million records / 1.6]
set -u
TAG=$1; M=${2:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 $R/tools/bam_e2e.py $M > /dev/null 2>&1
{
  for ENV in "" "SK_BAMFILE_THREADS=8" "SK_BAMFILE_THREADS=8 SK_BAMFILE_CHUNK_LOG2=24" "SK_BAMFILE_THREADS=12 SK_BAMFILE_CHUNK_LOG2=24" "SK_BAMFILE_THREADS=8 SK_BAMFILE_CHUNK_LOG2=23"; do
    echo "== $ENV"
    env $ENV SK_BAMFILE_TRACE=1 timeout -k 10 200 python3 $R/tools/r06/bam_file_info.py /dev/shm/sk_scale.bam 2>&1 | grep -v amdgpu.ids
  done
  TIMEFORMAT="  %R s wall  %U user  %S sys"
  echo "== sam statistics (CLI)"
  for i in 1 2 3; do time (SK_BAMFILE_TRACE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam); done
} 2>&1 | tee $OUT/bam_file.txt
rm -f /dev/shm/sk_scale.bam
