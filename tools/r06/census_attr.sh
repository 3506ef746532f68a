#!/bin/bash
# Round 6, verdict item 1: attribute the independent-rows census launch — kernel trace, PMC (traffic + SQ), phase stamps.
# usage (GPU box): bash tools/r06/census_attr.sh <tag>
set -u
TAG=${1:-r06_census_indep}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in noisy_indep clean_indep noisy; do
  W=/tmp/ct_$CASE; rm -rf $W
  rocprofv3 --kernel-trace --stats --output-format csv -d $W -- python3 $R/tools/census_one.py $CASE 32000000 4 > $W.log 2>&1
  tail -1 $W.log > $OUT/trace_$CASE.txt
  python3 - $W >> $OUT/trace_$CASE.txt <<'PY'
import csv, glob, os, sys
t = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(t)) if "census" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower() or "memset" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows[-16:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sk::", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:12.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {name}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))}")
PY
  cat $OUT/trace_$CASE.txt
done
# PMC of the independent noisy launch: traffic and SQ counters for all census kernels
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/cpi$i
  rocprofv3 --pmc $grp --kernel-include-regex "census" --output-format csv -d /tmp/cpi$i -- python3 $R/tools/census_one.py noisy_indep 32000000 3 > /tmp/cpi$i.log 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/cpi1 /tmp/cpi2 /tmp/cpi3 /tmp/cpi4 > $OUT/pmc_noisy_indep.txt 2>&1
cat $OUT/pmc_noisy_indep.txt
cd $R
SK_STAMPS_CASES=noisy,noisy_indep python3 tools/census_stamps.py 32000000 > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
