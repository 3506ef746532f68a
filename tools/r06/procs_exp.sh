#!/bin/bash
# `sam statistics` on the 3.6 GB file as N processes one after the other (does one process's exit make the next one wait?) and as repeated calls of
# one process.   usage (GPU box): bash tools/r06/procs_exp.sh [N]
set -u
N=${1:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 tools/bam_e2e.py 20 > /dev/null 2>&1
TIMEFORMAT="  %R s wall  %U user  %S sys"
for ENV in "" "SK_BAMFILE_NO_VMM=1"; do
  echo "== processes one after the other $ENV"
  for i in $(seq $N); do time (env $ENV SK_BAMFILE_TRACE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam 2>&1 | grep -E "waited" | cut -c1-120); done
  echo "== calls of one process $ENV"
  env $ENV BAM_INFO_REPS=6 timeout -k 10 300 python3 tools/r06/bam_file_info.py /dev/shm/sk_scale.bam 2>&1 | grep -v amdgpu.ids | sed 's/;.*read+copy/; read+copy/' | cut -c1-110
done
rm -f /dev/shm/sk_scale.bam
