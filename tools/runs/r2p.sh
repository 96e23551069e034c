set -x
mkdir -p gpurun_out/r2p
timeout -k 10 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_configs.py::test_config_quarter_degree > gpurun_out/r2p/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2p/pytest_gpu.log
tail -n 6 gpurun_out/r2p/pytest_gpu.log
python bench.py --no-cpu-baseline > gpurun_out/r2p/bench.log 2> gpurun_out/r2p/bench.err
tail -n 1 gpurun_out/r2p/bench.log | cut -c1-700
timeout -k 10 400 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 > gpurun_out/r2p/probe_quarter.log 2>&1
tail -n 1 gpurun_out/r2p/probe_quarter.log | cut -c1-900
