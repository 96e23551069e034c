set -x
mkdir -p gpurun_out/r2i
timeout -k 10 800 python -m pytest tests/test_gpu_dist.py tests/test_abi.py -q -m gpu -x > gpurun_out/r2i/pytest_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2i/pytest_dist.log
tail -n 25 gpurun_out/r2i/pytest_dist.log
