set -x
mkdir -p gpurun_out/r3t
for cfg in "centred isop" "donor isop" "centred const" "upwind3 const"; do
set -- $cfg
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv $1 --hmix $2 --max-iters 3000 > gpurun_out/r3t/probe_1deg_$1_$2.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r3t/probe_1deg_$1_$2.log").read().strip().splitlines()[-1])
print("$1 $2", "iters", d["iters"], "solve_s", d["solve_s"], "status", d["status"])
PY
done
for om in 1.0 1.2; do
NKP_ML_OMEGA=$om timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv centred --hmix isop --max-iters 3000 > gpurun_out/r3t/tmp.log 2>&1
tail -n 1 gpurun_out/r3t/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centred isop omega $om iters', d['iters'], d['solve_s'])"
done
NKP_PRECOND_STEPS=2 timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv centred --hmix isop --max-iters 3000 > gpurun_out/r3t/tmp.log 2>&1
tail -n 1 gpurun_out/r3t/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centred isop 2 cycles iters', d['iters'], d['solve_s'])"
NKP_KRYLOV=bicgstab timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv centred --hmix isop --krylov 1 --max-iters 3000 > gpurun_out/r3t/tmp.log 2>&1
tail -n 1 gpurun_out/r3t/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centred isop bicgstab iters', d['iters'], d['solve_s'], d['status'])"
timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 --adv centred --hmix isop --restart 400 --max-iters 3000 > gpurun_out/r3t/tmp.log 2>&1
tail -n 1 gpurun_out/r3t/tmp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('centred isop restart 400 iters', d['iters'], d['solve_s'], d['status'])"
