set -x
mkdir -p gpurun_out/r2r
for cfg in "2 3" "2 2" "2 1" "3 2" "3 1" "1 2"; do
set -- $cfg
NKP_ML_COARSE_FROM=$1 NKP_ML_SMOOTH_COARSE=$2 timeout -k 10 300 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2r/probe_1deg_cf$1_nc$2.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/r2r/probe_1deg_cf$1_nc$2.log").read().strip().splitlines()[-1])
print("coarse_from $1 nu_coarse $2", "cycle_ms", round(d["precond_ms"],3), "iters", d["iters"], "solve_s", d["solve_s"])
PY
done
