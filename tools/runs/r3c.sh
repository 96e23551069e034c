set -x
mkdir -p gpurun_out/r3c
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r3c/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c/pytest_gpu.log
tail -n 16 gpurun_out/r3c/pytest_gpu.log
