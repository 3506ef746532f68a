# how `sam statistics` scales with the inflate threads on this host — usage: bash tools/bam_scale.sh [million records]
TIMEFORMAT="  %R s wall  %U user  %S sys"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; grep -c processor /proc/cpuinfo
python3 - "$R" "${1:-20}" <<'PY'
import os, sys, subprocess, time
sys.path.insert(0, sys.argv[1])
os.environ["E2E_NO_ORACLE"] = "1"
os.environ["BAM_KEEP"] = "/dev/shm/sk_scale.bam"
PY
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 python3 $R/tools/bam_e2e.py ${1:-20} > /dev/null 2>&1
ls -la /dev/shm/sk_scale.bam
for t in 8 16 32 64 128; do
  echo "threads=$t"; time (SEQKIT_THREADS=$t $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam > /dev/null)
done
echo no-mmap; time (SEQKIT_NO_MMAP=1 SEQKIT_THREADS=64 $R/seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam > /dev/null)
echo gzip-dc; time (gzip -dc < /dev/shm/sk_scale.bam | head -c 1000000000 | wc -c)
rm -f /dev/shm/sk_scale.bam
