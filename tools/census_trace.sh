#!/bin/bash
# kernel trace of tools/census_rates.py: durations of the census kernels, one line per launch group (the last repetition of each case).
# usage: bash tools/census_trace.sh <tag> [rows]
set -u
TAG=$1; ROWS=${2:-32000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
W=/tmp/skprof_$TAG
rm -rf $W; mkdir -p $W $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/trace -- python3 "$R/tools/census_rates.py" $ROWS > $W/rates.txt 2>&1
grep "G rows/s" $W/rates.txt > $OUT/rates.txt
python3 - $W/trace > $OUT/kernels.txt <<'PY'
import csv, glob, os, sys
t = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(t)) if "census" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sk::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if name.startswith("census_kernel"): groups.append([])
    if groups: groups[-1].append((name, d))
for i, g in enumerate(groups):
    if i % 4 == 3: print(f"case {i // 4}: " + "  ".join(f"{n} {d:.1f} us" for n, d in g) + f"   total {sum(d for _, d in g):.1f} us")
PY
cat $OUT/kernels.txt; cut -c1-150 $OUT/rates.txt
