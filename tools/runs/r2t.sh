set -x
mkdir -p gpurun_out/r2t
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 > gpurun_out/r2t/probe_1deg.log 2>&1
grep -h "multilevel setup:" gpurun_out/r2t/probe_1deg.log | cut -c1-260
tail -n 1 gpurun_out/r2t/probe_1deg.log | cut -c1-420
timeout -k 10 500 python tools/probe_gpu.py --grid 1440x720x80 --restart 60 --verbose 1 > gpurun_out/r2t/probe_quarter.log 2>&1
grep -h "multilevel setup:" gpurun_out/r2t/probe_quarter.log | cut -c1-260
tail -n 1 gpurun_out/r2t/probe_quarter.log | cut -c1-900
