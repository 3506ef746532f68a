#!/bin/bash
# kernel durations of one census shape under a sweep of an environment knob.  usage: bash tools/census_sweep.sh <tag> <case> <VAR> <values...>
set -u
TAG=$1; CASE=$2; VAR=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  W=/tmp/sweep_$v; rm -rf $W
  env $VAR=$v true
  export $VAR=$v
  rocprofv3 --kernel-trace --output-format csv -d $W -- python3 "$R/tools/census_one.py" $CASE 32000000 5 > $W.log 2>&1
  python3 - $W "$VAR=$v" <<'PY' | tee -a $OUT/sweep.txt
import csv, glob, os, sys
t = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(t)) if "census" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sk::", "").split("<")[0].replace("census_", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if name == "kernel": groups.append([])
    if groups: groups[-1].append((name, d))
import statistics
g = groups[1:]
names = [n for n, _ in g[0]]
med = [statistics.median(x[i][1] for x in g) for i in range(len(names))]
print(sys.argv[2], "  ".join(f"{n} {m:.1f}" for n, m in zip(names, med)), f"  total {sum(med):.1f} us", flush=True)
PY
done
