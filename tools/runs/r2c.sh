set -x
mkdir -p gpurun_out/r2c
python -m pytest tests -q -m gpu > gpurun_out/r2c/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest_gpu.log
tail -n 3 gpurun_out/r2c/pytest_gpu.log
python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --verbose 1 --precond-steps 1 --max-iters 1500 > gpurun_out/r2c/probe_quarter_deg_k33.log 2>&1
tail -n 1 gpurun_out/r2c/probe_quarter_deg_k33.log
python tools/probe_gpu.py --grid 1440x720x80 --restart 100 --k33 0 --verbose 1 --precond-steps 1 --max-iters 1500 > gpurun_out/r2c/probe_quarter_deg_legacy.log 2>&1
tail -n 1 gpurun_out/r2c/probe_quarter_deg_legacy.log
