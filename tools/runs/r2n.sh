set -x
mkdir -p gpurun_out/r2n
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stream or column_blocks" > gpurun_out/r2n/pytest_stream.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2n/pytest_stream.log
tail -n 12 gpurun_out/r2n/pytest_stream.log
for st in 0 1; do
NKP_COLSTREAM=$st timeout -k 10 300 python - <<'PY'
import sys, os
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
for grid in ((320,384,60),(100,116,60)):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=0)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
    out=[os.environ.get('NKP_COLSTREAM'), grid]
    for w,k in ((3,'smoother_spmv_bytes'),(4,'column_solve_bytes'),(1,'cycle_bytes')):
        ms=s.time_kernel(w,reps=100); b=s.get_int(k); out.append((w, round(ms*1e3,1),'us', round(b/ms/1e9,2),'TB/s'))
    print(out)
    s.close()
PY
done
