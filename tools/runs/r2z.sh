set -x
mkdir -p gpurun_out/r2z
for lv in 80 64; do
NKP_COLSTREAM_LEVELS=$lv timeout -k 10 500 python - <<'PY'
import sys, os, time
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
import numpy as np
p = synth.generate(imt=1440, jmt=720, km=80, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
r = np.random.default_rng(1).standard_normal(p.flat_len)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, restart=8)
print("levels<=", os.environ["NKP_COLSTREAM_LEVELS"], "cycle_ms", round(s.time_kernel(1, reps=20), 2), "colsolve_us", round(s.time_kernel(4, reps=50)*1e3, 1), "spmv_half_us", round(s.time_kernel(3, reps=50)*1e3,1), flush=True)
z = s.precond_apply(r)
np.save(f"gpurun_out/r2z/z_{os.environ['NKP_COLSTREAM_LEVELS']}.npy", z[:2000000])
s.close()
PY
done
python -c "
import numpy as np
a=np.load('gpurun_out/r2z/z_80.npy'); b=np.load('gpurun_out/r2z/z_64.npy'); print('identical (first 2M):', bool(np.array_equal(a,b)))"
rm -f gpurun_out/r2z/z_*.npy
