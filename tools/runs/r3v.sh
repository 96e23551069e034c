set -x
mkdir -p gpurun_out/r3v
export NKP_BENCH_BACKEND=gloo
port=29720
for N in 2 4; do
for ras in 1 0; do
port=$((port+1))
NKP_DIST_RAS=$ras timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port bench.py --gpus $N --steps 1 --warmup 1 --multi-gpu strong > gpurun_out/r3v/bench_strong_N${N}_ras${ras}.log 2> gpurun_out/r3v/bench_strong_N${N}_ras${ras}.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3v/bench_strong_N${N}_ras${ras}.log").read().strip().splitlines()[-1])
print("bands $N ras $ras iterations", d["solve"]["iterations"], "ms", d["ms_per_step"], "setup", d["solve"]["setup_s"], d["solve"]["relres_checked_with_torch"])
PY
done
done
port=$((port+1))
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 1 --warmup 1 > gpurun_out/r3v/bench_c4_N2.log 2> gpurun_out/r3v/bench_c4_N2.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3v/bench_c4_N2.log").read().strip().splitlines()[-1])
print("c4 N2 iterations", d["solve"]["iterations"], "ms", d["ms_per_step"], "setup", d["solve"]["setup_s"], d["solve"]["relres_checked_with_torch"])
PY
