set -x
mkdir -p gpurun_out/r2h
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or restatement or fixed_linear" > gpurun_out/r2h/pytest_fused.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest_fused.log
tail -n 12 gpurun_out/r2h/pytest_fused.log
for f in 0 1; do
NKP_ML_FUSED=$f timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2h/probe_1deg_fused$f.log 2>&1
NKP_ML_FUSED=$f timeout -k 10 200 python tools/probe_gpu.py --grid 100x116x60 > gpurun_out/r2h/probe_3deg_fused$f.log 2>&1
done
tail -n 1 gpurun_out/r2h/probe*.log | cut -c1-1000
