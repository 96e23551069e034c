set -x
mkdir -p gpurun_out/r4u
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4u/pytest_gpu.log 2>&1
tail -n 5 gpurun_out/r4u/pytest_gpu.log
python bench.py > gpurun_out/r4u/bench.log 2> gpurun_out/r4u/bench.err || exit 1
tail -c 600 gpurun_out/r4u/bench.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4u/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --rhs-batch 0 --round1-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r4u/bench_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4u/prof -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/r4u/bench_kernel_stats.csv
find gpurun_out/r4u/prof -type f ! -name "*stats*" -delete
head -n 12 gpurun_out/r4u/bench_kernel_stats.csv
