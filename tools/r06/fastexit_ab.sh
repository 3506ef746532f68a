set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/r06/demux_prof.sh r06_z 2>&1 | grep -E "^==|process:|hip  " | cut -c1-200 | head -40
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 tools/bam_e2e.py 20 > /dev/null 2>&1
TIMEFORMAT="  %R s wall  %U user  %S sys"
for E in "" "SEQKIT_FAST_EXIT=1" "" "SEQKIT_FAST_EXIT=1"; do echo "== sam statistics $E"; time (env $E SK_BAMFILE_TRACE=1 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam 2>&1 | grep -E "waited"); done
rm -f /dev/shm/sk_scale.bam
