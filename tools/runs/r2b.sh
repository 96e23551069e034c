set -x
mkdir -p gpurun_out/r2b
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "restatement or fixed_linear or coupled_tracers_with_grid or chained" > gpurun_out/r2b/pytest_ml.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest_ml.log
for k33 in 0 1; do
  python tools/probe_gpu.py --grid 100x116x60 --refine 12 --k33 $k33 --verbose 1 --max-iters 3000 > gpurun_out/r2b/probe_3deg_ref12_k33_$k33.log 2>&1
  python tools/probe_gpu.py --grid 320x384x60 --k33 $k33 --verbose 1 --max-iters 3000 > gpurun_out/r2b/probe_1deg_k33_$k33.log 2>&1
done
NKP_ML_SPLIT=0 python tools/probe_gpu.py --grid 100x116x60 --refine 12 --k33 1 --max-iters 3000 > gpurun_out/r2b/probe_3deg_ref12_k33_1_nosplit.log 2>&1
NKP_ML_SPLIT=0 python tools/probe_gpu.py --grid 320x384x60 --k33 1 --max-iters 3000 > gpurun_out/r2b/probe_1deg_k33_1_nosplit.log 2>&1
python tools/probe_gpu.py --grid 640x768x60 --k33 1 --verbose 1 --max-iters 3000 > gpurun_out/r2b/probe_halfdeg_k33_1.log 2>&1
tail -n 1 gpurun_out/r2b/*.log
