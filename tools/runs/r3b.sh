set -x
mkdir -p gpurun_out/r3b
for cfg in "1 0 64" "2 0 64" "1 1 64" "2 1 64"; do
set -- $cfg
NKP_COL_W3=$1 NKP_COLSTREAM_W3=$2 NKP_COLSTREAM_LEVELS=$3 timeout -k 10 300 python - <<'PY'
import sys, os
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
out=[os.environ.get('NKP_COL_W3'), os.environ.get('NKP_COLSTREAM_W3')]
for grid in ((320,384,60),(100,116,60)):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=0)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
    out.append((grid[0], 'cycle_us', round(s.time_kernel(1,reps=100)*1e3,1), 'col_us', round(s.time_kernel(4,reps=100)*1e3,1)))
    s.close()
print(out)
PY
done
for cfg in "1 1 80" "1 0 80"; do
set -- $cfg
NKP_COL_W3=$1 NKP_COLSTREAM_W3=$2 NKP_COLSTREAM_LEVELS=$3 timeout -k 10 500 python - <<'PY'
import sys, os
sys.path.insert(0,'.')
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=1440, jmt=720, km=80, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, restart=8)
print("0.25deg stream80 w3=", os.environ["NKP_COLSTREAM_W3"], "cycle_ms", round(s.time_kernel(1, reps=20), 2), "colsolve_us", round(s.time_kernel(4, reps=50)*1e3, 1), flush=True)
PY
done
