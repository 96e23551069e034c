mkdir -p gpurun_out/r4f
for v in 2 1; do
NKP_COL_LDSRES=$v NKP_COLSTREAM_MIN=20000 timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --rhs-batch 0 --round1-steps 0 > gpurun_out/r4f/bench_ldsres$v.log 2>gpurun_out/r4f/bench_ldsres$v.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r4f/bench_ldsres$v.log").read().strip().splitlines()[-1])
print("LDSRES=$v ms_per_step", d["ms_per_step"], [ (k["kernel"][:28], round(k["avg_launch_ms"]*1e3,2), round(k["frac"],3)) for k in d["roofline"]["kernels"]])
PY
done
