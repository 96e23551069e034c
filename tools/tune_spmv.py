#!/usr/bin/env python3
"""Developer probe: same-process A/B of SpMV launch parameters (env knobs read per launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, precond=solver.PRECOND_COLUMN_JACOBI, restart=4)
nbytes = s.get_int("spmv_bytes")
for rep in range(3):
    for w in (4, 6, 8, 12, 16, 24, 32, 48, 64):
        os.environ["NKP_SPMV_WGS"] = str(w)
        ms = s.time_kernel(0, reps=100)
        print(f"rep {rep} wgs/CU {w:3d}: {ms*1e3:7.1f} us  {nbytes/ms/1e6:7.0f} GB/s", flush=True)
