#!/usr/bin/env python3
"""Developer probe: synthetic circulation file -> bin/gen_A -> matrix file -> GPU solve.
Reports gen_A wall time, matrix size, iterations and solve time (not part of the test suite)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nk_ocn_tracer_jacobian_precond_amd import circ, nc3, solver

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
ap.add_argument("--adv", default="upwind3")
ap.add_argument("--hmix", default="isop_file")
ap.add_argument("--vmix", default="file")
ap.add_argument("--sink", default="const_shallow 365.0 10.0e2")
ap.add_argument("--divfree", type=int, default=1)
ap.add_argument("--restart", type=int, default=200)
ap.add_argument("--max-iters", type=int, default=5000)
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--precond", type=int, default=3)
ap.add_argument("--verbose", type=int, default=0)
ap.add_argument("--keep", default="")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
d = a.keep or tempfile.mkdtemp(prefix="gen_A_probe_")
os.makedirs(d, exist_ok=True)
res = dict(grid=a.grid, adv=a.adv, hmix=a.hmix, vmix=a.vmix)

t0 = time.time()
F, fills = circ.make_circulation(imt, jmt, km, seed=0)
circ.write_circ_file(os.path.join(d, "circ.nc"), F, fills, nc_type="float32")
res["circ_s"] = round(time.time() - t0, 2)
res["circ_MB"] = round(os.path.getsize(os.path.join(d, "circ.nc")) / 1e6, 1)
with open(os.path.join(d, "gen_A.opt"), "w") as fh:
    fh.write(f"circ_fname {d}/circ.nc\nadv_type {a.adv}\nl_adv_enforce_divfree {a.divfree}\nhmix_type {a.hmix}\nvmix_type {a.vmix}\nsink_type {a.sink}\n")
t0 = time.time()
r = subprocess.run([os.path.join(ROOT, "nk_ocn_tracer_jacobian_precond_amd", "bin", "gen_A"), "-o", os.path.join(d, "gen_A.opt"),
                    os.path.join(d, "matrix.nc")], capture_output=True, text=True)
res["gen_A_s"] = round(time.time() - t0, 2)
if r.returncode:
    print(r.stderr)
    sys.exit(1)

m = nc3.NcFile(os.path.join(d, "matrix.nc"))
rp, ci, val = m.get("rowptr"), m.get("colind"), m.get("nzval_row_wise")
ii, jj, kk = (m.get(f"tracer_state_ind_to_{c}") for c in "ijk")
n = len(rp) - 1
res.update(n=n, nnz=len(val))
rows = np.repeat(np.arange(n), np.diff(rp))
diag = np.zeros(n)
diag[rows[ci == rows]] = val[ci == rows]
off = np.bincount(rows, weights=np.abs(val) * (ci != rows), minlength=n)
res["diag_dominant_rows"] = float(np.mean(np.abs(diag) >= off))
res["positive_offdiag_frac"] = float(np.mean(val[ci != rows] * np.sign(-diag[rows[ci != rows]]) > 0))

col_start = np.concatenate([np.flatnonzero(kk == 0), [len(kk)]]).astype(np.int32)
blk = solver.column_blocks(col_start, len(kk), 1)
cci, ccj = solver.column_coords(ii, jj, col_start, 1)
t0 = time.time()
s = solver.NkpSolver(rp, ci, val, blk, col_i=cci, col_j=ccj, restart=a.restart, max_iters=a.max_iters, rtol=a.rtol, precond=a.precond,
                     verbose=a.verbose)
res["setup_s"] = round(time.time() - t0, 3)
res["levels"] = s.get_int("levels")
b = np.random.default_rng(1).standard_normal(n)
t0 = time.time()
x, info = s.solve(b, raise_on_fail=False)
res["solve_s"] = round(time.time() - t0, 3)
res.update(info)
print(json.dumps(res))
