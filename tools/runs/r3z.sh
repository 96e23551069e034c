set -x
mkdir -p gpurun_out/r3z
# second use of this script: the same run with NKP_DIST_RAS=0 (the first one wrote bench_c5_N2.log)
export NKP_BENCH_BACKEND=gloo
( while true; do sleep 45; echo "alive $(date +%T) $(free -g | awk '/Mem/{print $3}') GB used" >> gpurun_out/r3z/heartbeat.log; done ) &
HB=$!
NKP_DIST_RAS=0 timeout -k 10 1000 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29762 bench.py --gpus 2 --steps 1 --warmup 1 --multi-gpu strong --grid 1440x720x80 --restart 60 > gpurun_out/r3z/bench_c5_N2_noras.log 2> gpurun_out/r3z/bench_c5_N2_noras.err
rc=$?
kill $HB
tail -n 5 gpurun_out/r3z/bench_c5_N2_noras.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3z/bench_c5_N2_noras.log").read().strip().splitlines()[-1])
print("c5 bands 2 iterations", d["solve"]["iterations"], "ms", d["ms_per_step"], "setup", d["solve"]["setup_s"], d["solve"]["relres_checked_with_torch"], d["solve"]["distributed"])
PY
exit $rc
