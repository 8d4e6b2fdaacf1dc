# Counter groups for the consume pass on the nerf_lego table and on S1, one rocprofv3 --pmc pass per group, each under its own
# timeout (the TA_BUSY / TA_*_STALLED group aborted the profiler and hung the box call for 20 minutes: left out).
for w in LEGO S1; do
i=0
for g in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
         "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum" \
         "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"; do
  i=$((i+1)); echo "=== $w group $i"; timeout 180 bash tools/pmc.sh pmc_${w}_$i "$g" tools/bwd_only.py $w 3 2>&1 | grep -A12 "bin_consume"
done; done
