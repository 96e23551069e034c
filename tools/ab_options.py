#!/usr/bin/env python3
"""Developer probe: iteration counts for option combinations on one grid (same process, same rhs)."""
import argparse, itertools, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
b = np.random.default_rng(1).standard_normal(p.flat_len)
for mlf32, bf32, reorth, restart in [(0, 0, 1, 100), (0, 0, 1, 200), (0, 0, 0, 200), (1, 0, 0, 200), (0, 1, 0, 200), (1, 1, 0, 200), (1, 1, 1, 200)]:
    os.environ["NKP_ML_F32"] = str(mlf32)
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, basis_f32=bf32, reorth=reorth, restart=restart, max_iters=4000)
    t0 = time.time(); x, info = s.solve(b, raise_on_fail=False); dt = time.time() - t0
    print(f"ml_f32={mlf32} basis_f32={bf32} reorth={reorth} restart={restart}: iters {info['iters']} relres {info['relres']:.2e} solve {dt:.3f}s", flush=True)
    s.close()
