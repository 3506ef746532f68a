cd ${GRAFT_REPO_ROOT:-.}
BAM_KEEP=/dev/shm/sk_big.bam E2E_NO_ORACLE=1 timeout -k 10 900 python3 tools/bam_e2e.py 100 > /dev/null 2>&1
TIMEFORMAT="  %R s wall"
sleep 8
for i in 1 2 3; do time (SK_BAMFILE_TRACE=1 seqkit_amd/bin/sam statistics /dev/shm/sk_big.bam 2>&1 | grep -E "sk_bam_file_reduce:" | cut -c1-120); done
sleep 10
time (SK_BAMFILE_TRACE=1 seqkit_amd/bin/sam statistics /dev/shm/sk_big.bam 2>&1 | grep -E "sk_bam_file_reduce:" | cut -c1-120)
rm -f /dev/shm/sk_big.bam
