mkdir -p gpurun_out/r4i
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_resident" > gpurun_out/r4i/pytest_kernel.log 2>&1 || { tail -n 40 gpurun_out/r4i/pytest_kernel.log; exit 1; }
NKP_LDSRES_EARLY=0 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_resident" >> gpurun_out/r4i/pytest_kernel.log 2>&1 || { tail -n 40 gpurun_out/r4i/pytest_kernel.log; exit 1; }
tail -n 2 gpurun_out/r4i/pytest_kernel.log
for rep in 1 2; do
for v in 1 0; do
NKP_LDSRES_EARLY=$v timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --rhs-batch 0 --round1-steps 0 > gpurun_out/r4i/bench_early$v.log 2>gpurun_out/r4i/bench_early$v.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r4i/bench_early$v.log").read().strip().splitlines()[-1])
print("EARLY=$v ms_per_step", round(d["ms_per_step"],2), [ (k["kernel"][:28], round(k["avg_launch_ms"]*1e3,2), round(k["frac"],3)) for k in d["roofline"]["kernels"]])
PY
done
done
