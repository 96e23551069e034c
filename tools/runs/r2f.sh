set -x
mkdir -p gpurun_out/r2f
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r2f/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest_gpu.log
tail -n 30 gpurun_out/r2f/pytest_gpu.log
for eq in 0 1; do
NKP_EQUIL=$eq python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2f/probe_1deg_equil$eq.log 2>&1
NKP_EQUIL=$eq python tools/probe_gpu.py --grid 320x384x60 --k33 0 > gpurun_out/r2f/probe_1deg_legacy_equil$eq.log 2>&1
done
tail -n 1 gpurun_out/r2f/probe*.log
