#!/usr/bin/env python3
"""Developer check: at sizes where 32-bit index arithmetic could bite (0.25 degree: nnz ~ 0.9e9), compare the
library's SpMV and one multilevel cycle's linearity against torch's own sparse CSR product (64-bit indices)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nk_ocn_tracer_jacobian_precond_amd import solver, synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="1440x720x80")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
t0 = time.time()
p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0)
print(f"generated n={p.flat_len} nnz={p.nnz} in {time.time() - t0:.0f}s", flush=True)
blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, restart=10)
print("solver built", flush=True)
n = p.flat_len
x = torch.randn(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
s.spmv_device(x.data_ptr(), y.data_ptr())
A = torch.sparse_csr_tensor(torch.from_numpy(p.rowptr.astype(np.int64)).cuda(), torch.from_numpy(p.colind.astype(np.int64)).cuda(),
                            torch.from_numpy(p.nzval).cuda(), size=(n, n))
yr = (A @ x.unsqueeze(1)).squeeze(1)
err = float(torch.linalg.norm(y - yr) / torch.linalg.norm(yr))
bad = int((torch.abs(y - yr) > 1e-9 * torch.abs(yr).max()).sum())
print(f"spmv relative difference vs torch: {err:.3e}, rows off by more than 1e-9 of max: {bad}")
r1 = np.random.default_rng(0).standard_normal(n)
r2 = np.random.default_rng(1).standard_normal(n)
z1, z2, z3 = s.precond_apply(r1), s.precond_apply(r2), s.precond_apply(2.0 * r1 - r2)
print("cycle linearity:", np.linalg.norm(z3 - (2.0 * z1 - z2)) / np.linalg.norm(z3), " |z|/|r| =", np.linalg.norm(z1) / np.linalg.norm(r1),
      " zero rows in z:", int((z1 == 0.0).sum()), " nonfinite:", int((~np.isfinite(z1)).sum()))
