#!/usr/bin/env python3
"""Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes): launches the SpMV of the
1 degree bench workload plus a 16 B/lane calibration stream (scale_to_kernel inside one Krylov step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, precond=solver.PRECOND_COLUMN_JACOBI, restart=8)
print("n", p.flat_len, "nnz", p.nnz, "spmv_bytes", s.get_int("spmv_bytes"))
print("spmv_ms", s.time_kernel(0, reps=40))
print("step_ms", s.time_kernel(2, reps=5, arg=3))
