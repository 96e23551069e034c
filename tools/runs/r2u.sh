set -x
mkdir -p gpurun_out/r2u
for cfg in "centred isop" "donor isop" "centred const" "upwind3 const"; do
set -- $cfg
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv $1 --hmix $2 --max-iters 3000 > gpurun_out/r2u/probe_1deg_$1_$2.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2u/probe_1deg_$1_$2.log").read().strip().splitlines()[-1])
print("$1 $2", "iters", d["iters"], "solve_s", d["solve_s"], "status", d["status"], "relres", d["relres"])
PY
done
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --no-geo 1 --max-iters 3000 > gpurun_out/r2u/probe_1deg_nogeo.log 2>&1
tail -n 1 gpurun_out/r2u/probe_1deg_nogeo.log | cut -c1-200; tail -n 1 gpurun_out/r2u/probe_1deg_nogeo.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no-geo iters',d['iters'],'solve_s',d['solve_s'])"
timeout -k 10 600 python tools/probe_gpu.py --grid 1440x720x80 --restart 200 > gpurun_out/r2u/probe_quarter_m200.log 2>&1
tail -n 1 gpurun_out/r2u/probe_quarter_m200.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('0.25 deg m=200 iters',d['iters'],'solve_s',d['solve_s'],'setup',d['setup_s'])"
