#!/bin/bash
# sk_bam_file_reduce and `sam statistics` on a BAM of <M> million records (BASELINE config 5 names 200 M records; /dev/shm has to hold the file).
# usage (GPU box): bash tools/r06/bam_big.sh [M]
set -u
M=${1:-100}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
df -h /dev/shm | tail -1
BAM_KEEP=/dev/shm/sk_big.bam E2E_NO_ORACLE=1 timeout -k 10 900 python3 tools/bam_e2e.py $M > /dev/null 2>&1
ls -la /dev/shm/sk_big.bam || exit 0
BAM_INFO_REPS=3 timeout -k 10 600 python3 tools/r06/bam_file_info.py /dev/shm/sk_big.bam 2>&1 | grep -v amdgpu.ids
TIMEFORMAT="  %R s wall  %U user  %S sys"
time (SK_BAMFILE_TRACE=1 seqkit_amd/bin/sam statistics /dev/shm/sk_big.bam)
time (SEQKIT_HOST_INFLATE=1 seqkit_amd/bin/sam statistics /dev/shm/sk_big.bam)
rm -f /dev/shm/sk_big.bam
