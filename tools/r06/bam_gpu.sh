#!/bin/bash
# `sam statistics` / `sam fragment lengths` on the 32 M-record BAM of tools/bam_scale.sh: the device-inflate path against the host-inflate path.
# usage (GPU box): bash tools/r06/bam_gpu.sh <out dir> [million records / 1.6]
set -u
OUT=$1; M=${2:-20}
R=${GRAFT_REPO_ROOT:-$(pwd)}
TIMEFORMAT="  %R s wall  %U user  %S sys"
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 $R/tools/bam_e2e.py $M > /dev/null 2>&1
ls -la /dev/shm/sk_scale.bam
{
  echo "== sam statistics, device inflate (default)"
  for i in 1 2 3; do time (SK_BAMFILE_TRACE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam); done
  echo "== sam statistics, host inflate (SEQKIT_HOST_INFLATE=1)"
  for i in 1 2; do time (SEQKIT_HOST_INFLATE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam); done
  echo "== sam fragment lengths, device inflate: md5 of the output against the host path's"
  time ($R/seqkit_amd/bin/sam fragment lengths /dev/shm/sk_scale.bam | md5sum)
  time (SEQKIT_HOST_INFLATE=1 $R/seqkit_amd/bin/sam fragment lengths /dev/shm/sk_scale.bam | md5sum)
  echo "== the stages (tools/r06/bam_file_info.py)"
  python3 $R/tools/r06/bam_file_info.py /dev/shm/sk_scale.bam
} 2>&1 | tee $OUT/bam_gpu.txt
rm -f /dev/shm/sk_scale.bam
