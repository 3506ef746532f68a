#!/bin/bash
# the device inflater's tests and rates, nothing else.  usage (GPU box): bash tools/r06/inflate_quick.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -12 | tee $OUT/inflate_tests.txt
timeout -k 10 300 python3 tools/r06/inflate_rate.py 1024 2>&1 | grep -v amdgpu.ids | tee $OUT/inflate_rate.txt
