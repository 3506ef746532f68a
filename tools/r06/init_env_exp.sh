cd ${GRAFT_REPO_ROOT:-.}
BAM_KEEP=/dev/shm/sk_scale.bam E2E_NO_ORACLE=1 timeout -k 10 600 python3 tools/bam_e2e.py 20 > /dev/null 2>&1
TIMEFORMAT="%R s wall"
for E in "" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=1" "HSA_ENABLE_SDMA=0" "AMD_LOG_LEVEL=0" "HIP_VISIBLE_DEVICES=0" "ROCR_VISIBLE_DEVICES=0" "SEQKIT_CTXS_PER_GPU=1" "HSA_NO_SCRATCH_RECLAIM=1"; do
  for i in 1 2 3; do echo -n "$E: "; ( time (env $E SK_BAMFILE_TRACE=1 seqkit_amd/bin/sam statistics /dev/shm/sk_scale.bam 2>&1 | grep -E "waited" | cut -c17-100) ) 2>&1 | tr "\n" " "; echo; done
done
rm -f /dev/shm/sk_scale.bam
