set -x
mkdir -p gpurun_out/r2o
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "stream or column_blocks or restatement" > gpurun_out/r2o/pytest_stream.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o/pytest_stream.log
tail -n 5 gpurun_out/r2o/pytest_stream.log
for cfg in "0 0 64" "1 50000 64" "1 50000 32" "1 20000 64" "1 20000 32" "1 5000 32"; do
set -- $cfg
NKP_COLSTREAM=$1 NKP_COLSTREAM_MIN=$2 NKP_COLSTREAM_GW=$3 timeout -k 10 300 python - <<'PY'
import sys, os
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
for grid in ((320,384,60),):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=0)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
    out=[os.environ.get('NKP_COLSTREAM'), os.environ.get('NKP_COLSTREAM_MIN'), os.environ.get('NKP_COLSTREAM_GW')]
    for w,k in ((4,'column_solve_bytes'),(1,'cycle_bytes')):
        ms=s.time_kernel(w,reps=100); out.append((w, round(ms*1e3,1),'us'))
    print(out)
    s.close()
PY
done
