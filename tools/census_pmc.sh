#!/bin/bash
# PMC counters of the census front kernel on one shape: LDS activity / conflicts / instruction mix / waits, separate passes (8 SQ slots each).
# usage: bash tools/census_pmc.sh <tag> [case] [rows]
set -u
TAG=$1; CASE=${2:-noisy}; ROWS=${3:-32000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD"; do
  i=$((i+1)); rm -rf /tmp/cp$i
  rocprofv3 --pmc $grp --kernel-include-regex "census_kernel" --output-format csv -d /tmp/cp$i -- python3 $R/tools/census_one.py $CASE $ROWS 3 > /tmp/cp$i.log 2>&1
  tail -2 /tmp/cp$i.log
done
python3 $R/tools/pmc_summary.py /tmp/cp1 /tmp/cp2 /tmp/cp3 > $OUT/pmc_$CASE.txt
cat $OUT/pmc_$CASE.txt
