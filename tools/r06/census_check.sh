#!/bin/bash
# Round 6: the census after a change — its GPU tests, then every shape's launch time (and reset time), then a kernel trace of two shapes.
# usage (GPU box): bash tools/r06/census_check.sh <tag> [pytest -k expression | none]
set -u
TAG=${1:-r06_census_check}; KEXPR=${2:-census}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
if [ "$KEXPR" != "none" ]; then
  timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" 2>&1 | tail -15 > $OUT/tests.txt
  cat $OUT/tests.txt
fi
for CASE in noisy_indep clean_indep noisy clean exact sub distinct; do
  python3 tools/census_one.py $CASE 32000000 5 2>&1 | tail -1
done | tee $OUT/rates.txt
cd /tmp && export TMPDIR=/tmp
for CASE in noisy_indep clean_indep; do
  W=/tmp/ct_$CASE; rm -rf $W
  rocprofv3 --kernel-trace --output-format csv -d $W -- python3 $R/tools/census_one.py $CASE 32000000 4 > $W.log 2>&1
  python3 - $W > $OUT/trace_$CASE.txt <<'PY'
import csv, glob, os, sys
t = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(t)) if "census" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
for r in rows[-12:]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sk::", "")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:12.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us  {name}")
PY
  echo "== $CASE"; cat $OUT/trace_$CASE.txt
done
