set -x
mkdir -p gpurun_out/r3h
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r3h/pytest_dist.log 2>&1; tail -n 6 gpurun_out/r3h/pytest_dist.log
NKP_DIST_OVERLAP=0 timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -q -m gpu -x -k "distributed_solve" > gpurun_out/r3h/pytest_dist_nooverlap.log 2>&1; tail -n 3 gpurun_out/r3h/pytest_dist_nooverlap.log
export NKP_BENCH_BACKEND=gloo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --steps 1 --warmup 1 --multi-gpu strong > gpurun_out/r3h/bench_strong_N2.log 2> gpurun_out/r3h/bench_strong_N2.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3h/bench_strong_N2.log").read().strip().splitlines()[-1])
print("bands 2", d["solve"]["iterations"], d["solve"]["relres_checked_with_torch"])
PY
grep -h "interior" gpurun_out/r3h/*.err gpurun_out/r3h/*.log | head -3
