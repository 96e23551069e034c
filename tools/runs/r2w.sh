set -x
mkdir -p gpurun_out/r2w
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2w/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2w/pytest_gpu.log
tail -n 6 gpurun_out/r2w/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2w/smoke.log 2>&1; tail -n 2 gpurun_out/r2w/smoke.log
python bench.py > gpurun_out/r2w/bench.log 2> gpurun_out/r2w/bench.err
tail -n 1 gpurun_out/r2w/bench.log | cut -c1-400
