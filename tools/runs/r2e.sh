set -x
mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_configs.py -q -m gpu -x --durations=0 > gpurun_out/r2e/pytest_configs.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2e/pytest_configs.log
tail -n 15 gpurun_out/r2e/pytest_configs.log
python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --k33 1 --precond-steps 2 --max-iters 1500 > gpurun_out/r2e/probe_quarter_deg_k33_steps2.log 2>&1
tail -n 1 gpurun_out/r2e/probe_quarter_deg_k33_steps2.log
python tools/probe_gpu.py --grid 320x384x60 --tracers 4 --restart 100 --verbose 1 > gpurun_out/r2e/probe_1deg_4tracers.log 2>&1
tail -n 1 gpurun_out/r2e/probe_1deg_4tracers.log
