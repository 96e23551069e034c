mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_resident or stream_kernel or wave_per_column or fused or tail" > gpurun_out/r4g/pytest_kernel.log 2>&1 || { tail -n 40 gpurun_out/r4g/pytest_kernel.log; exit 1; }
tail -n 2 gpurun_out/r4g/pytest_kernel.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --rhs-batch 0 --round1-steps 0 > gpurun_out/r4g/bench.log 2>gpurun_out/r4g/bench.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/r4g/bench.log").read().strip().splitlines()[-1])
print("default ms_per_step", d["ms_per_step"], d["solve"]["iterations"], [ (k["kernel"][:28], round(k["avg_launch_ms"]*1e3,2), round(k["frac"],3)) for k in d["roofline"]["kernels"]])
PY
for g in 100x116x60 640x768x60; do
timeout -k 10 300 python tools/probe_gpu.py --grid $g > gpurun_out/r4g/tmp.log 2>&1 || exit 1
tail -n 1 gpurun_out/r4g/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', 'cycle_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
done
