set -x
mkdir -p gpurun_out/r3y
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r3y/pytest_dist.log 2>&1 || { tail -n 40 gpurun_out/r3y/pytest_dist.log; exit 1; }
tail -n 3 gpurun_out/r3y/pytest_dist.log
export NKP_BENCH_BACKEND=gloo
port=29740
for N in 2 4; do
port=$((port+1))
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 1 --warmup 1 > gpurun_out/r3y/bench_c4_cell_N${N}.log 2> gpurun_out/r3y/bench_c4_cell_N${N}.err || { tail -n 30 gpurun_out/r3y/bench_c4_cell_N${N}.err; exit 1; }
python - <<PY
import json
d=json.loads(open("gpurun_out/r3y/bench_c4_cell_N${N}.log").read().strip().splitlines()[-1])
print("c4 cell-major N $N iterations", d["solve"]["iterations"], "ms", d["ms_per_step"], "setup", d["solve"]["setup_s"], d["solve"]["relres_checked_with_torch"], d["solve"]["krylov_iteration_ms"])
PY
done
