set -x
mkdir -p gpurun_out/r2l
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r2l/pytest_parity.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest_parity.log
tail -n 8 gpurun_out/r2l/pytest_parity.log
timeout -k 10 200 python tools/probe_gpu.py --grid 320x384x60 > gpurun_out/r2l/probe_1deg.log 2>&1
timeout -k 10 200 python tools/probe_gpu.py --grid 100x116x60 > gpurun_out/r2l/probe_3deg.log 2>&1
python - <<'PY'
import sys
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
for w,k in ((3,'smoother_spmv_bytes'),(4,'column_solve_bytes'),(1,'cycle_bytes')):
    ms=s.time_kernel(w,reps=100); b=s.get_int(k); print(w, round(ms*1e3,1),'us', round(b/ms/1e9,2),'TB/s')
PY
tail -q -n 1 gpurun_out/r2l/probe*.log | cut -c1-900
