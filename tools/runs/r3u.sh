set -x
mkdir -p gpurun_out/r3u
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q > gpurun_out/r3u/pytest_dist.log 2>&1
tail -n 15 gpurun_out/r3u/pytest_dist.log
