set -x
mkdir -p gpurun_out/r2m
export NKP_BENCH_BACKEND=gloo
for N in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 2 --warmup 1 > gpurun_out/r2m/bench_c4_N$N.log 2> gpurun_out/r2m/bench_c4_N$N.err
tail -n 1 gpurun_out/r2m/bench_c4_N$N.log | cut -c1-1400
tail -n 3 gpurun_out/r2m/bench_c4_N$N.err
done
