set -x
mkdir -p gpurun_out/r2y
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stream" > gpurun_out/r2y/pytest.log 2>&1; tail -n 3 gpurun_out/r2y/pytest.log
timeout -k 10 900 python - <<'PY'
import sys, os, time
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=1440, jmt=720, km=80, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
import numpy as np
r = np.random.default_rng(1).standard_normal(p.flat_len)
zs = {}
for lv in ("64", "80"):
    os.environ["NKP_COLSTREAM_LEVELS"] = lv
    # the env is read once per process (static): use the min-columns knob instead to switch it off for the first pass
    os.environ["NKP_COLSTREAM"] = "0" if lv == "64" else "1"
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, restart=8)
    print("stream", os.environ["NKP_COLSTREAM"], "cycle_ms", round(s.time_kernel(1, reps=20), 2), "colsolve_us", round(s.time_kernel(4, reps=50)*1e3, 1), flush=True)
    zs[lv] = s.precond_apply(r)
    s.close()
print("bit identical:", bool(np.array_equal(zs["64"], zs["80"])))
PY
