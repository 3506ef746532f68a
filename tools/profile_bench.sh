#!/bin/bash
# Kernel trace + PMC passes of bench.py on the GPU box; writes compact summaries to gpurun_out/<tag>/.
# usage: bash tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-prof}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
W=/tmp/skprof_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" 2>/dev/null | tail -1 > $OUT/bench.json
# the traced run keeps its own JSON line: its HIP-event kernel time and the trace's durations of its timed steps come from
# the SAME process (placements — and with them the kernel time — differ from process to process)
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -- python3 $R/bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-extra "$@" > $W/trace.log 2>$W/trace.err
grep '^{"metric"' $W/trace.log | tail -1 > $OUT/traced_bench.json
python3 $R/tools/summarize_prof.py $W/trace $OUT/rocprof $OUT/traced_bench.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
  "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "sk::" --output-format csv -d $W/pmc$i -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra "$@" > $W/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $W/pmc* > $OUT/pmc_summary.txt 2>&1
cat $OUT/pmc_summary.txt
LAYOUT=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['layout'])")
PAIRS=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['config']['clusters_per_gpu'])")
KERN=$(python3 -c "import json;print(json.load(open('$OUT/bench.json'))['roofline']['kernel'])")
python3 $R/tools/make_pmc_traffic.py $OUT/pmc_summary.txt $PAIRS $LAYOUT "$KERN" > $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json
cut -c1-400 $OUT/bench.json
