# Regenerates the raw material of profiles/ on the GPU box (run through gpurun): bash tools/refresh_profiles.sh [tag]
# then, back in the repo: python tools/make_profiles.py gpurun_out/<tag>f <tag>
set -x
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}f
mkdir -p $OUT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
tail -c 600 $OUT/bench_line.json
python tools/gpu_check.py > $OUT/parity_and_timing.txt 2>&1
python tools/config_c.py > $OUT/config_c.txt 2>&1
python tools/tiled_check.py > $OUT/tiled_check.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --psnr-steps 0 --nerf-steps 0 --no-secondary > $OUT/stats_line.json 2>/dev/null
for op in fwd bwd; do for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_${op}_${c} -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py $op -1 3 3 > /dev/null 2>&1
done; done
# forward kernels, unit counters (separate passes; SQ: 8 slots, TCC: 4 slots): cell-sorted path and the unsorted kernels
for mode in tiled plain; do
if [ $mode = plain ]; then export SHACIRA_OPTIONS="tiled=0"; else unset SHACIRA_OPTIONS; fi
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/ctr_${mode}_sq -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ctr_${mode}_ta -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d $OUT/ctr_${mode}_tcp -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/ctr_${mode}_tcc -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
done
unset SHACIRA_OPTIONS
ls $OUT
