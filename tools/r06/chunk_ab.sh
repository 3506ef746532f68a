#!/bin/bash
# census combine chunk A/B: tools/ab/census_c<chunk>.so against the product build, four shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for LIB in "" tools/ab/census_c8192.so tools/ab/census_c4096.so tools/ab/census_c2048.so; do
  echo "== lib: ${LIB:-product (16384)}"
  for CASE in noisy_indep clean_indep noisy sub; do
    SK_LIB=$LIB timeout -k 10 100 python3 tools/census_one.py $CASE 32000000 5 2>&1 | tail -1
  done
done
