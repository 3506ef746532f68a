#!/bin/bash
# Kernel trace + PMC passes of any python tool on the GPU box; writes compact summaries to gpurun_out/<tag>/.
# usage: bash tools/profile_cmd.sh <tag> <kernel regex> <script.py> [args...]
set -u
TAG=$1; RE=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
W=/tmp/skprof_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -- python3 "$R/$1" "${@:2}" > $W/trace.log 2>&1
python3 $R/tools/summarize_prof.py $W/trace $OUT/rocprof
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "$RE" --output-format csv -d $W/pmc$i -- python3 "$R/$1" "${@:2}" > $W/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $W/pmc* > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
