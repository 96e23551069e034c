#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes): launches the two kernels the 1 degree
V-cycle spends most of its time in -- the smoother's residual rows of one colour of the fine level and the water-column
solves of that colour -- plus a 16 B/lane calibration stream (scale_to_kernel inside one Krylov step of a second solver
whose kernels carry other template arguments, so the names do not mix)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
cal = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, precond=solver.PRECOND_COLUMN_JACOBI, restart=8)
cal.time_kernel(2, reps=3, arg=3)
cal.close()
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj)
out = dict(n=p.flat_len, nnz=p.nnz, smoother_spmv_bytes=s.get_int("smoother_spmv_bytes"), column_solve_bytes=s.get_int("column_solve_bytes"),
           smoother_ms=s.time_kernel(3, reps=10), column_ms=s.time_kernel(4, reps=10))
print(json.dumps(out))
