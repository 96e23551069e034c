set -x
mkdir -p gpurun_out/r2x
timeout -k 10 400 python bench.py --force-dist --steps 2 --warmup 1 --no-cpu-baseline --rhs-batch 0 > gpurun_out/r2x/bench_forcedist_rccl.log 2> gpurun_out/r2x/bench_forcedist_rccl.err
tail -n 1 gpurun_out/r2x/bench_forcedist_rccl.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['solve']['iterations'], d['config']['multi_gpu'][-80:])"
tail -n 3 gpurun_out/r2x/bench_forcedist_rccl.err
timeout -k 10 400 python bench.py --force-dist --comm torch --steps 2 --warmup 1 --no-cpu-baseline --rhs-batch 0 > gpurun_out/r2x/bench_forcedist_torch.log 2> gpurun_out/r2x/bench_forcedist_torch.err
tail -n 1 gpurun_out/r2x/bench_forcedist_torch.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['solve']['iterations'], d['config']['multi_gpu'][-80:])"
