set -x
mkdir -p gpurun_out/r3i
timeout -k 10 1100 python -m pytest tests -q -m gpu --deselect tests/test_gpu_configs.py::test_config_quarter_degree > gpurun_out/r3i/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i/pytest_gpu.log
tail -n 12 gpurun_out/r3i/pytest_gpu.log
