set -x
mkdir -p gpurun_out/r3o
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r3o/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3o/pytest_gpu.log
tail -n 8 gpurun_out/r3o/pytest_gpu.log
python bench.py > gpurun_out/r3o/bench.log 2> gpurun_out/r3o/bench.err
tail -n 1 gpurun_out/r3o/bench.log | cut -c1-300
