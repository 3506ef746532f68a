#!/bin/bash
# Everything profiles/ holds for a round, in ONE session on the GPU box: bench line + kernel trace + PMC passes of the headline,
# the N = 2 same-device line, the CLI's kernel trace, PMC of the lookup kernel and of trim alone, the rates of every tool, the
# census kernels' trace and PMC, the GPU test run.  usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash tools/profile_bench.sh $TAG > $OUT/profile_bench.log 2>&1
SK_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --pairs 4000000 --steps 5 --warmup 2 --cpu-sample 0 --no-extra --placements 1 > $OUT/two_ranks_same_device.json 2> $OUT/two_ranks.err
bash tools/profile_cli_demux.sh $TAG 2000000 > $OUT/cli_demux.log 2>&1
bash tools/profile_cmd.sh ${TAG}_lut_dual "demux_lut" tools/demux_one.py dual 10000000 > $OUT/lut_dual_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_lut_cfg3 "demux_lut" tools/demux_one.py cfg3 10000000 > $OUT/lut_cfg3_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_trim_uniform "tile_pass_kernel" tools/trim_one.py uniform 16000000 > $OUT/trim_uniform_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_fused_single "tile_pass_kernel" tools/fused_one.py single 16000000 > $OUT/fused_single_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_lut_384 "demux_lut" tools/demux_one.py dual384 10000000 > $OUT/lut_384_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_lut_dual_100m "demux_lut" tools/demux_one.py dual 100000000 > $OUT/lut_dual_100m_pmc.log 2>&1
bash tools/profile_cmd.sh ${TAG}_lut_cfg3_100m "demux_lut" tools/demux_one.py cfg3 100000000 > $OUT/lut_cfg3_100m_pmc.log 2>&1
python3 tools/lut_cold_ab.py --sheets 16,96,384,1000 cur 2>&1 | grep -v amdgpu.ids > $OUT/lut_cold.txt
python3 tools/lut_cold_ab.py --detail --sheets 16,96 cur 2>&1 | grep -v amdgpu.ids >> $OUT/lut_cold.txt
bash tools/profile_cmd.sh ${TAG}_census_noisy "census_" tools/census_one.py noisy 32000000 3 > $OUT/census_noisy_pmc.log 2>&1
bash tools/census_trace.sh ${TAG}_census > $OUT/census_trace.log 2>&1
( cd seqkit_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSK_CENSUS_STAMPS -o ../../tools/ab/census_stamps.so sk_kernels.hip sk_census.hip sk_inflate.hip sk_deflate.hip sk_capi.hip sk_bamfile.cpp sk_lut.cpp -ldl -lz -pthread ) > $OUT/census_stamps_build.log 2>&1
SK_STAMPS_CASES=exact,clean,noisy,noisy_indep python3 tools/census_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/census_stamps.txt
bash tools/r06/census_attr.sh ${TAG}_census_indep > $OUT/census_indep.log 2>&1
# (round 6) where the inflater's waves spend their cycles: the -DSK_INF_STAMPS build
( cd seqkit_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSK_INF_STAMPS -o ../../tools/ab/inf_stamps.so sk_kernels.hip sk_census.hip sk_inflate.hip sk_deflate.hip sk_capi.hip sk_bamfile.cpp sk_lut.cpp -ldl -lz -pthread ) > $OUT/inf_stamps_build.log 2>&1
python3 tools/r06/inflate_rate.py 1024 tools/ab/inf_stamps.so 2>&1 | grep -v amdgpu.ids > $OUT/inflate_stamps.txt
# (round 6) every secondary kernel whose fraction is quoted: a kernel trace and the PMC passes (tools/r06/kernel_one.py)
for K in mask bam fragments sequence152 sequence148 inflate_random inflate_sorted deflate; do
  bash tools/profile_cmd.sh ${TAG}_k_$K "sk::" tools/r06/kernel_one.py $K > $OUT/k_$K.log 2>&1
done
python3 tools/r06/lut_repro.py 100000000 4 2>&1 | grep -v amdgpu.ids > $OUT/lut_repro.txt
python3 tools/r06/many_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/many_rate.txt
python3 tools/r06/inflate_rate.py 1024 2>&1 | grep -v amdgpu.ids > $OUT/inflate_rate.txt
python3 tools/r06/deflate_rate.py 256 2>&1 | grep -v amdgpu.ids > $OUT/deflate_rate.txt
bash tools/r06/bam_gpu.sh $OUT > $OUT/bam_gpu.log 2>&1
bash tools/r06/deflate_e2e.sh $OUT 80 > $OUT/deflate_e2e.log 2>&1
DEMUX_PROF_QUIET=1 bash tools/r06/demux_prof.sh $TAG 80 > $OUT/demux_prof.log 2>&1
{ tools/ab/insert_exp; } > $OUT/insert_exp.txt 2>&1
{
  echo "== tools/rates.py"; python3 tools/rates.py 2>&1 | grep -v amdgpu.ids
  echo "== tools/demux_ab.py (DEMUX_DETAIL=1, forms default / table in the vector cache / no table)"
  DEMUX_DETAIL=1 DEMUX_FORMS="default;SK_DEMUX_LDSTAB=0;SK_NO_HASH_DEMUX=1" DEMUX_N=1000000,10000000,100000000 python3 tools/demux_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== tools/trim_exp.py"; python3 tools/trim_exp.py 16000000 2>&1 | grep -v amdgpu.ids
  echo "== tools/census_rates.py"; python3 tools/census_rates.py 2>&1 | grep -v amdgpu.ids
  echo "== tools/seq_ab.py (pitch 152, 148; the LDS-tile kernel, then SK_SEQ_TILE=0)"; python3 tools/seq_ab.py 2>&1 | grep -v amdgpu.ids; SEQ_STRIDE=148 python3 tools/seq_ab.py 2>&1 | grep -v amdgpu.ids
  SK_SEQ_TILE=0 SEQ_STRIDE=148 python3 tools/seq_ab.py 2>&1 | grep -v amdgpu.ids
} > $OUT/rates.txt 2>&1
{ echo "== tools/bam_scale.sh 20"; bash tools/bam_scale.sh 20 2>&1; echo "== tools/bam_paths.sh 20"; bash tools/bam_paths.sh 20 2>&1; } > $OUT/bam_host.txt 2>&1
timeout -k 10 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -15 > $OUT/gpu_tests.txt
ls -la $OUT
