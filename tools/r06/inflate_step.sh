#!/bin/bash
# Round 6: the device inflater after a change — its tests, its rate on both kinds of stream, and sk_bam_file_reduce on the 3.6 GB file
# under the pipeline's knobs.   usage (GPU box): bash tools/r06/inflate_step.sh <tag> [million records / 1.6]
set -u
TAG=$1; M=${2:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -8 | tee $OUT/inflate_tests.txt
timeout -k 10 300 python3 tools/r06/inflate_rate.py 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/inflate_rate.txt
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 $R/tools/bam_e2e.py $M > /dev/null 2>&1
{
  for ENV in "" "SK_BAMFILE_ROUNDS=0 SK_BAMFILE_STREAMS=1 SK_BAMFILE_BATCH=8192" "SK_BAMFILE_STREAMS=1" "SK_BAMFILE_ROUNDS=0" "SK_BAMFILE_THREADS=8" "SK_BAMFILE_THREADS=12" "SK_BAMFILE_THREADS=8 SK_BAMFILE_CHUNK_LOG2=24"; do
    echo "== $ENV"
    env $ENV timeout -k 10 200 python3 $R/tools/r06/bam_file_info.py /dev/shm/sk_scale.bam 2>&1 | grep -v amdgpu.ids
  done
  TIMEFORMAT="  %R s wall  %U user  %S sys"
  echo "== sam statistics (CLI)"
  for i in 1 2; do time (SK_BAMFILE_TRACE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam); done
} 2>&1 | tee $OUT/bam_file.txt
rm -f /dev/shm/sk_scale.bam
