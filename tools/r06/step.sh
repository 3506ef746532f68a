#!/bin/bash
# Round 6: one GPU session of checks, every step under its own timeout (a hung kernel must not hold the box).
# usage (GPU box): bash tools/r06/step.sh <tag> <steps...>     steps: inflate census many rates bamfile clitests
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for STEP in "$@"; do
  echo "=== $STEP"
  case $STEP in
    inflate) timeout -k 10 300 python3 -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/inflate_tests.txt ;;
    census) timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k census 2>&1 | tail -15 | tee $OUT/census_tests.txt ;;
    many) timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_batches or by_table_into" 2>&1 | tail -15 | tee $OUT/many_tests.txt ;;
    clitests) timeout -k 10 900 python3 -m pytest tests/test_cli_gpu.py -x -q -m gpu -k "sam" 2>&1 | tail -15 | tee $OUT/cli_sam_tests.txt ;;
    rates)
      for CASE in noisy_indep clean_indep noisy clean exact sub distinct; do
        timeout -k 10 100 python3 tools/census_one.py $CASE 32000000 5 2>&1 | tail -1
      done | tee $OUT/census_rates.txt ;;
    trace)
      cd /tmp && export TMPDIR=/tmp
      for CASE in noisy_indep clean_indep; do
        W=/tmp/ct_$CASE; rm -rf $W
        timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $W -- python3 $R/tools/census_one.py $CASE 32000000 4 > $W.log 2>&1
        python3 $R/tools/r06/trace_tail.py $W census fill > $OUT/trace_$CASE.txt 2>&1
        echo "== $CASE"; cat $OUT/trace_$CASE.txt
      done
      cd $R ;;
    bamfile) timeout -k 10 900 bash tools/r06/bam_gpu.sh $OUT 2>&1 | tail -40 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
