# A/B of two builds of the library on trim shapes: kernel-trace durations (not HIP events) — usage: bash tools/pmc_ab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" tools/ab/base.so; do
  for kind in readlike5 readlike uniform; do
    rm -rf /tmp/pm; SK_LIB=${lib:+$R/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -- python3 $R/tools/trim_one.py $kind 16000000 > /tmp/pm.log 2>&1
    echo "== lib=${lib:-cur} kind=$kind $(grep -h tile_pass /tmp/pm/*/*kernel_stats.csv | cut -d, -f2-7 | tail -1)"
  done
done
done
