#!/usr/bin/env python3
"""Developer probe: kernel timings and iteration counts on one GPU (not part of the test suite)."""
import argparse
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
ap.add_argument("--adv", default="upwind3")
ap.add_argument("--hmix", default="isop")
ap.add_argument("--restart", type=int, default=200)
ap.add_argument("--max-iters", type=int, default=20000)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--precond", type=int, default=3)
ap.add_argument("--krylov", type=int, default=0)
ap.add_argument("--reorth", type=int, default=0)
ap.add_argument("--solve", type=int, default=1)
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--no-geo", type=int, default=0)
ap.add_argument("--tracers", type=int, default=1)
ap.add_argument("--ml-smooth", type=int, default=3)
ap.add_argument("--ml-levels", type=int, default=0)
ap.add_argument("--min-cos", type=float, default=0.3)
ap.add_argument("--refine", type=float, default=1.0, help="cell-level coefficients of a grid this many times finer (u x F, ah x F^2)")
ap.add_argument("--k33", type=int, default=1, help="isop: include the K33 vertical term of the Redi tensor")
ap.add_argument("--precond-steps", type=int, default=0)
ap.add_argument("--basis-f32", type=int, default=0)
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
t0 = time.time()
p = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, min_cos=a.min_cos, coupled_tracer_cnt=a.tracers,
                   u_scale=3.0 * a.refine, ah=4.0e6 * a.refine ** 2, isop_k33=bool(a.k33))
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, a.tracers)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), a.tracers)
tgen = time.time() - t0
t0 = time.time()
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=None if a.no_geo else ci, col_j=None if a.no_geo else cj, coupled_tracer_cnt=a.tracers, restart=a.restart, max_iters=a.max_iters, rtol=a.rtol,
                     precond=a.precond, krylov=a.krylov, reorth=a.reorth, verbose=a.verbose, ml_smooth=a.ml_smooth, ml_levels=a.ml_levels, precond_steps=a.precond_steps, basis_f32=a.basis_f32)
tsetup = time.time() - t0
res = dict(grid=a.grid, refine=a.refine, k33=a.k33, adv=a.adv, hmix=a.hmix, precond=a.precond, levels=s.get_int("levels"), ml_rows=s.get_int("ml_rows"), ml_nnz=s.get_int("ml_nnz"), n=p.flat_len, nnz=p.nnz, gen_s=round(tgen, 2), setup_s=round(tsetup, 3))
ms = s.time_kernel(0, reps=50)
res["spmv_ms"] = ms
res["spmv_GBs"] = s.get_int("spmv_bytes") / ms / 1e6
ms = s.time_kernel(1, reps=50)
res["precond_ms"] = ms
res["precond_GBs"] = s.get_int("precond_bytes") / ms / 1e6
for pos in (0, a.restart // 2, a.restart - 1):
    res[f"arnoldi_ms_j{pos}"] = s.time_kernel(2, reps=10, arg=pos)
if a.solve:
    b = np.random.default_rng(1).standard_normal(p.flat_len)
    t0 = time.time()
    x, info = s.solve(b, raise_on_fail=False)
    res["solve_s"] = round(time.time() - t0, 3)
    res.update(info)
print(json.dumps(res))
