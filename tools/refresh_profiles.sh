set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01f
python bench.py > gpurun_out/r01f/bench_line.json 2> gpurun_out/r01f/bench_err.txt
tail -c 600 gpurun_out/r01f/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01f/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --psnr-steps 0 --nerf-steps 0 --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r01f/stats_line.json 2>/dev/null
for op in fwd bwd; do for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r01f/pmc_${op}_${c} -- python3 $GRAFT_REPO_ROOT/tools/fwd_only.py $op -1 3 3 > /dev/null 2>&1
done; done
ls $GRAFT_REPO_ROOT/gpurun_out/r01f/*
