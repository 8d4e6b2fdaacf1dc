# Regenerates the raw material of profiles/ on the GPU box (run through gpurun): bash tools/refresh_profiles.sh [tag]
# then, back in the repo: python tools/make_profiles.py gpurun_out/<tag>f <tag>
set -x
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}f
mkdir -p $OUT
python -c "from bench import kernel_source_hash; print(kernel_source_hash())" > $OUT/source_hash.txt   # the sources measured
STAGES=${STAGES:-"bench checks stats pmc ctr bctr steps mfma"}    # subset to re-run, e.g. STAGES="bench ctr"
has() { case " $STAGES " in *" $1 "*) return 0;; esac; return 1; }
if has bench; then
( time timeout 600 python bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt ) 2> $OUT/bench_wall.txt
tail -c 600 $OUT/bench_line.json; cat $OUT/bench_wall.txt
fi
if has checks; then
timeout 300 python tools/attic/gpu_check.py > $OUT/parity_and_timing.txt 2>&1
timeout 300 python tools/attic/config_c.py > $OUT/config_c.txt 2>&1
timeout 300 python tools/attic/tiled_check.py > $OUT/tiled_check.txt 2>&1
fi
cd /tmp && export TMPDIR=/tmp
if has stats; then
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --psnr-steps 0 --nerf-steps 0 --no-secondary > $OUT/stats_line.json 2>/dev/null
fi
if has pmc; then
# the PLANNED pair (forward builds the plan, backward reads it: what bench.py's step runs), three calls; kernels are attributed to
# their operator by name (tools/make_profiles.py)
for c in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_pair_${c} -- python3 $GRAFT_REPO_ROOT/tools/plan_prof.py 3 > /dev/null 2>&1
done
fi
if has ctr; then
# forward kernels, unit counters (separate passes; SQ: 8 slots, TCC: 4 slots). Keep the TA / TCP groups at two counters:
# the five-counter TA group (with the *_STALLED_BY_* pair) hung the profiler twice on this pool: cell-sorted path and the unsorted kernels
for mode in tiled plain; do
if [ $mode = plain ]; then export SHACIRA_OPTIONS="tiled=0"; else unset SHACIRA_OPTIONS; fi
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ctr_${mode}_sq -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum --kernel-trace --output-format csv -d $OUT/ctr_${mode}_ta -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $OUT/ctr_${mode}_tcp -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/ctr_${mode}_tcc -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py fwd -1 3 3 > /dev/null 2>&1
done
unset SHACIRA_OPTIONS
fi
if has bctr; then
# backward kernels, unit counters (same groups; separate passes)
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ctr_bwd_sq -- python3 $GRAFT_REPO_ROOT/tools/plan_prof.py 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/ctr_bwd_lds -- python3 $GRAFT_REPO_ROOT/tools/plan_prof.py 3 > /dev/null 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/ctr_bwd_tcc -- python3 $GRAFT_REPO_ROOT/tools/plan_prof.py 3 > /dev/null 2>&1
fi
if has steps; then
# what the operators sit inside: one eager step of the NeRF fit / the image fit (kernel time vs wall time), and the graphed image fit
for w in nerf nerf_pool nerf_graphed image image_graphed; do
timeout 300 python3 $GRAFT_REPO_ROOT/tools/attic/step_breakdown.py $w 1000 > $OUT/step_${w}_plain.json 2>/dev/null   # the unprofiled wall time
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/step_$w -- python3 $GRAFT_REPO_ROOT/tools/attic/step_breakdown.py $w 1000 > $OUT/step_$w.json 2>/dev/null
find $OUT/step_$w -name "*kernel_trace.csv" -delete    # 10^5 rows each: only the stats summary is used (gpurun_out merges <= 64 MiB)
done
fi
if has mfma; then
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/ctr_mlp128 -- python3 $GRAFT_REPO_ROOT/tools/attic/mlp128_check.py > $OUT/mlp128_check.txt 2>&1
fi
du -sh $OUT
ls $OUT
