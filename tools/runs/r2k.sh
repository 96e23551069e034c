set -x
mkdir -p gpurun_out/r2k
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tail or restatement or fixed_linear or clones" > gpurun_out/r2k/pytest_tail.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2k/pytest_tail.log
tail -n 12 gpurun_out/r2k/pytest_tail.log
for rows in 0 8000 16000 40000 100000; do
NKP_ML_TAIL_ROWS=$rows timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 > gpurun_out/r2k/probe_1deg_tail$rows.log 2>&1
NKP_ML_TAIL_ROWS=$rows timeout -k 10 200 python tools/probe_gpu.py --grid 100x116x60 > gpurun_out/r2k/probe_3deg_tail$rows.log 2>&1
done
grep -h "single-workgroup" gpurun_out/r2k/probe_1deg_tail*.log
tail -q -n 1 gpurun_out/r2k/probe*.log | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['grid'], 'precond_ms', round(d['precond_ms'],3), 'iters', d['iters'], 'solve_s', d['solve_s'])"
