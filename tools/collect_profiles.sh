#!/bin/bash
# gpurun_out/<tag>* (what tools/profile_round.sh wrote on the GPU box) -> profiles/<tag>_*: the summaries that are kept.
# usage: bash tools/collect_profiles.sh r04
set -eu
T=${1:-r06}; G=gpurun_out; P=profiles
for f in bench.json traced_bench.json pmc_summary.txt pmc_traffic.json rocprof_kernel_stats.csv rocprof_kernel_trace_head.csv rocprof_timed_steps.txt \
         two_ranks_same_device.json cli_demux_kernel_stats.csv cli_demux_kernel_trace_head.csv rates.txt bam_host.txt gpu_tests.txt insert_exp.txt lut_cold.txt lut_repro.txt many_rate.txt inflate_rate.txt inflate_stamps.txt deflate_rate.txt bam_gpu.txt deflate_e2e.txt demux_prof.txt; do cp $G/$T/$f $P/${T}_$f; done
cp $G/$T/pmc_traffic.json $P/pmc_traffic.json
grep -v amdgpu.ids $G/$T/census_stamps.txt > $P/${T}_census_stamps.txt
cp $G/${T}_census/kernels.txt $P/${T}_census_kernels.txt
for k in census_noisy fused_single lut_384 lut_cfg3 lut_dual lut_cfg3_100m lut_dual_100m trim_uniform; do
  cp $G/${T}_$k/pmc_summary.txt $P/${T}_${k}_pmc_summary.txt
  cp $G/${T}_$k/rocprof_kernel_stats.csv $P/${T}_${k}_rocprof_kernel_stats.csv
  cp $G/${T}_$k/rocprof_kernel_trace_head.csv $P/${T}_${k}_rocprof_kernel_trace_head.csv
done
for k in mask bam fragments sequence152 sequence148 inflate_random inflate_sorted deflate; do
  cp $G/${T}_k_$k/pmc_summary.txt $P/${T}_k_${k}_pmc_summary.txt
  cp $G/${T}_k_$k/rocprof_kernel_stats.csv $P/${T}_k_${k}_rocprof_kernel_stats.csv
done
for f in trace_noisy_indep.txt trace_clean_indep.txt pmc_noisy_indep.txt stamps.txt; do cp $G/${T}_census_indep/$f $P/${T}_census_indep_$f; done
