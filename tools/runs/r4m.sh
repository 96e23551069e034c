mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -k self_test > gpurun_out/r4m/pytest_dist.log 2>&1 || { tail -n 40 gpurun_out/r4m/pytest_dist.log; exit 1; }
tail -n 3 gpurun_out/r4m/pytest_dist.log
