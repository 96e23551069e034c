set -x
mkdir -p gpurun_out/r3g
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r3g/pytest.log 2>&1; tail -n 4 gpurun_out/r3g/pytest.log
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 > gpurun_out/r3g/probe_1deg.log 2>&1
grep -h "multilevel setup:\|nkp_create:" gpurun_out/r3g/probe_1deg.log | cut -c1-330
tail -n 1 gpurun_out/r3g/probe_1deg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'])"
timeout -k 10 600 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --verbose 1 > gpurun_out/r3g/probe_quarter.log 2>&1
grep -h "multilevel setup:\|nkp_create:" gpurun_out/r3g/probe_quarter.log | cut -c1-330
tail -n 1 gpurun_out/r3g/probe_quarter.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'], 'cycle_ms', d['precond_ms'])"
