mkdir -p gpurun_out/r4v
export NKP_BENCH_BACKEND=gloo
( while true; do sleep 45; echo "alive $(date +%T) $(free -g | awk '/Mem/{print $3}') GB used" >> gpurun_out/r4v/heartbeat.log; done ) &
HB=$!
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29781 bench.py --gpus 5 --steps 1 --warmup 1 > gpurun_out/r4v/bench_c4_cell_N5.log 2> gpurun_out/r4v/bench_c4_cell_N5.err
rc=$?
kill $HB
tail -n 3 gpurun_out/r4v/bench_c4_cell_N5.err | cut -c1-300
python - <<PY
import json
d=json.loads(open("gpurun_out/r4v/bench_c4_cell_N5.log").read().strip().splitlines()[-1])
print("c4 cell-major N 5 iterations", d["solve"]["iterations"], "setup", d["solve"]["setup_s"], d["solve"]["relres_checked_with_torch"], d["solve"]["distributed"])
PY
exit $rc
