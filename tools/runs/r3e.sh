set -x
mkdir -p gpurun_out/r3e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "dense_inverse or restatement or coupled_tracers" > gpurun_out/r3e/pytest.log 2>&1; tail -n 8 gpurun_out/r3e/pytest.log
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 > gpurun_out/r3e/probe_1deg.log 2>&1
grep -h "multilevel setup:\|nkp_create:" gpurun_out/r3e/probe_1deg.log | cut -c1-420
tail -n 1 gpurun_out/r3e/probe_1deg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'], 'iters', d['iters'], 'solve_s', d['solve_s'])"
