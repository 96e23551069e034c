set -x
mkdir -p gpurun_out/r2d
python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --k33 1 --verbose 1 --precond-steps 1 --max-iters 1500 > gpurun_out/r2d/probe_quarter_deg_k33.log 2>&1
tail -n 1 gpurun_out/r2d/probe_quarter_deg_k33.log
for st in 1 2; do
python tools/probe_gpu.py --grid 320x384x60 --k33 1 --precond-steps $st > gpurun_out/r2d/probe_1deg_k33_steps$st.log 2>&1
python tools/probe_gpu.py --grid 320x384x60 --k33 0 --precond-steps $st > gpurun_out/r2d/probe_1deg_legacy_steps$st.log 2>&1
python tools/probe_gpu.py --grid 100x116x60 --k33 1 --precond-steps $st > gpurun_out/r2d/probe_3deg_k33_steps$st.log 2>&1
done
tail -n 1 gpurun_out/r2d/probe_1deg*.log gpurun_out/r2d/probe_3deg*.log
