set -x
mkdir -p gpurun_out/r2s
export NKP_BENCH_BACKEND=gloo
for N in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2961$N bench.py --gpus $N --steps 1 --warmup 1 --multi-gpu strong > gpurun_out/r2s/bench_strong_N$N.log 2> gpurun_out/r2s/bench_strong_N$N.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2s/bench_strong_N$N.log").read().strip().splitlines()[-1])
print("bands $N", d["solve"]["iterations"], d["solve"]["relres_checked_with_torch"], d["config"]["multi_gpu"][:60])
PY
done
