#!/usr/bin/env python3
"""Gram-Schmidt cost of one Arnoldi step at 1 degree: step time at basis positions 0 / 100 / 199 (nkp_time_kernel 2)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
with solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj) as s:
    t = {j: s.time_kernel(2, reps=10, arg=j) for j in (0, 100, 199)}
    n = p.flat_len
    gs100 = t[100] - t[0]
    print(json.dumps(dict(step_ms=t, gs_ms_at_100=gs100, gs_TBs=(2 * 100 + 3) * n * 8 / gs100 / 1e9)))
