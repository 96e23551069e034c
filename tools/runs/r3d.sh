set -x
mkdir -p gpurun_out/r3d
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --verbose 1 --solve 0 > gpurun_out/r3d/probe_1deg.log 2>&1
grep -h "multilevel setup:\|nkp_create:" gpurun_out/r3d/probe_1deg.log | cut -c1-420
tail -n 1 gpurun_out/r3d/probe_1deg.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('setup_s', d['setup_s'])"
