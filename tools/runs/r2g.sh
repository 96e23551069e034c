set -x
mkdir -p gpurun_out/r2g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r2g/prof -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --rhs-batch 0 > gpurun_out/r2g/bench_prof.log 2>&1
tail -n 2 gpurun_out/r2g/bench_prof.log | cut -c1-600
find gpurun_out/r2g/prof -name "*kernel_stats*" | head
for nu in 1 2; do
python tools/probe_gpu.py --grid 320x384x60 --ml-smooth $nu > gpurun_out/r2g/probe_1deg_nu$nu.log 2>&1
done
NKP_ML_BIG_FROM=2 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2g/probe_1deg_big2.log 2>&1
NKP_ML_BIG_FROM=1 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2g/probe_1deg_big1.log 2>&1
python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --ml-smooth 2 > gpurun_out/r2g/probe_quarter_nu2.log 2>&1
tail -n 1 gpurun_out/r2g/probe*.log | cut -c1-900
