set -x
mkdir -p gpurun_out/r3f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r3f/pytest.log 2>&1; tail -n 4 gpurun_out/r3f/pytest.log
timeout -k 10 600 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --verbose 1 > gpurun_out/r3f/probe_quarter.log 2>&1
grep -h "multilevel setup:\|nkp_create:" gpurun_out/r3f/probe_quarter.log | cut -c1-420
tail -n 1 gpurun_out/r3f/probe_quarter.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'cycle_ms', d['precond_ms'], 'spmv_ms', d['spmv_ms'])"
