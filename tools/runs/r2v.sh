set -x
mkdir -p gpurun_out/r2v
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "wave_per_column or restatement or identical" > gpurun_out/r2v/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2v/pytest.log
tail -n 6 gpurun_out/r2v/pytest.log
for wm in 0 2048 8192 30000 200000; do
NKP_COLWAVE_MAX=$wm timeout -k 10 300 python - <<'PY'
import sys, os
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
out=[os.environ.get('NKP_COLWAVE_MAX')]
for grid in ((320,384,60),(100,116,60)):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=0)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
    out.append((grid[0], 'cycle_us', round(s.time_kernel(1,reps=100)*1e3,1)))
    s.close()
print(out)
PY
done
