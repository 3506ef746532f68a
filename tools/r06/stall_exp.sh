set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 tools/bam_e2e.py 20 > /dev/null 2>&1
for F in ${STALL_FACTORS:-6 2}; do
  echo "== SK_BAMFILE_OUT_FACTOR=$F"
  SK_BAMFILE_OUT_FACTOR=$F BAM_INFO_REPS=9 timeout -k 10 300 python3 tools/r06/bam_file_info.py /dev/shm/sk_scale.bam 2>&1 | grep -v amdgpu.ids | sed 's/;.*read+copy/; read+copy/' | cut -c1-120
done
rm -f /dev/shm/sk_scale.bam
