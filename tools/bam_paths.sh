# which inflate path costs what on this host — usage: bash tools/bam_paths.sh [million records]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TIMEFORMAT="  %R s wall  %U user  %S sys"
ldconfig -p 2>/dev/null | grep -i deflate; ls /usr/lib/x86_64-linux-gnu/ 2>/dev/null | grep -i deflate
BAM_KEEP=/dev/shm/sk_paths.bam E2E_NO_ORACLE=1 python3 $R/tools/bam_e2e.py ${1:-20} > /dev/null 2>&1
for env in "" "SEQKIT_NO_LIBDEFLATE=1" "SEQKIT_NO_LIBDEFLATE=1 SEQKIT_ZLIB_INFLATE=1"; do
  echo "env: ${env:-default}"; time (env $env SEQKIT_THREADS=16 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_paths.bam > /dev/null)
done
rm -f /dev/shm/sk_paths.bam
