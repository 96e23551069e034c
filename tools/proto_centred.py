#!/usr/bin/env python3
"""CPU prototype (scipy, tests/ml_reference.py): iteration counts of cycle variants on CENTRED-advection operators (the
reference's default adv_type, src/gen_A.c:99).  Not part of the product or the tests.
  base      the product's cycle: every level smooths the low-order twin L
  fineA     level 0 smooths A itself (residuals and column blocks of A), coarse correction from the twin's hierarchy
  blend T   level 0 smooths (1 - T) A + T L
  inner K   preconditioner = K FGMRES steps on A preconditioned by the cycle (one-level K-cycle)
"""
import argparse, os, sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ml_reference as mlr
from nk_ocn_tracer_jacobian_precond_amd import synth
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
ap.add_argument("--adv", default="centred")
ap.add_argument("--hmix", default="isop")
ap.add_argument("--refine", type=float, default=1.0)
ap.add_argument("--variants", default="base,fineA,blend0.5,inner2")
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--nu", type=int, default=3)
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, u_scale=3.0 * a.refine, ah=4.0e6 * a.refine ** 2)
A = p.scipy_csr()
n = A.shape[0]
colid = np.cumsum(p.ind_k == 0) - 1
t0 = time.time()
levels = mlr.build(A, p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64), colid, nu=a.nu)
print(f"n={n} levels={[lv.n for lv in levels]} build {time.time()-t0:.1f}s", flush=True)
b = np.random.default_rng(1).standard_normal(n)


def fine_colours(M):
    lv = levels[0]
    C = M.tocoo()
    same = colid[C.row] == colid[C.col]
    Bd = sp.csr_matrix((C.data[same], (C.row[same], C.col[same])), shape=M.shape)
    colour = (lv.ci + lv.cj) % 2
    out = []
    for c in range(2):
        rows = np.flatnonzero(colour == c)
        out.append((rows, spla.splu(Bd[rows][:, rows].tocsc()), M[rows]))
    return out


def run(name, prec):
    its = [0]
    res = []
    M = spla.LinearOperator(A.shape, matvec=prec, dtype=np.float64)
    t0 = time.time()
    x, info = spla.gmres(A, b, M=M, rtol=a.rtol, restart=200, maxiter=3, callback=lambda r: (its.__setitem__(0, its[0] + 1), res.append(r)), callback_type="pr_norm")
    rr = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    print(f"{name:12s} its={its[0]:4d} relres={rr:.1e} time={time.time()-t0:.0f}s", flush=True)


saved = (levels[0].A, levels[0].colours)
for v in a.variants.split(","):
    levels[0].A, levels[0].colours = saved
    if v == "base":
        run(v, lambda r: mlr.cycle(levels, 0, np.asarray(r, np.float64), nu=a.nu))
    elif v == "fineA" or v.startswith("blend"):
        T = float(v[5:]) if v.startswith("blend") else 0.0
        M = ((1.0 - T) * A + T * saved[0]).tocsr()
        levels[0].A, levels[0].colours = M, fine_colours(M)
        run(v, lambda r: mlr.cycle(levels, 0, np.asarray(r, np.float64), nu=a.nu))
    elif v.startswith("inner"):
        K = int(v[5:])
        def prec(r, K=K):
            Mc = spla.LinearOperator(A.shape, matvec=lambda q: mlr.cycle(levels, 0, np.asarray(q, np.float64), nu=a.nu), dtype=np.float64)
            x, _ = spla.gmres(A, r, M=Mc, rtol=1e-30, restart=K, maxiter=1)
            return x
        run(v, prec)
