#!/usr/bin/env python3
"""CPU prototype (scipy): how should the multilevel preconditioner be split over P row-block ranks?
Counts FGMRES iterations for
  local      rank-local hierarchies only (what the distributed solver did first)
  add        local + additive global coarse correction from level LC of the global hierarchy
  mult       local, then a global coarse correction on the updated residual
Not part of the product or the tests."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ml_reference as mr
from nk_ocn_tracer_jacobian_precond_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
ap.add_argument("--P", type=int, default=4)
ap.add_argument("--mode", default="local")
ap.add_argument("--lc", type=int, default=2)
ap.add_argument("--rtol", type=float, default=1e-8)
ap.add_argument("--restart", type=int, default=200)
ap.add_argument("--maxit", type=int, default=3000)
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0)
A = p.scipy_csr()
n = A.shape[0]
cs = p.col_start()
colid = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
ci, cj, ck = p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64)
t0 = time.time()

# row-block partition snapped to column boundaries
cuts = [0]
for r in range(1, a.P):
    tgt = n * r // a.P
    cuts.append(int(cs[np.argmin(np.abs(cs - tgt))]))
cuts.append(n)
locs = []
for r in range(a.P):
    f, e = cuts[r], cuts[r + 1]
    Ab = A[f:e][:, f:e].tocsr()
    cid = colid[f:e] - colid[f]
    locs.append((f, e, mr.build(Ab, ci[f:e], cj[f:e], ck[f:e], cid)))
glob = None
if a.mode != "local" or a.P == 1:
    glob = mr.build(A, ci, cj, ck, colid)
    Pc = None
    for l in range(a.lc):
        Pc = glob[l].P if Pc is None else (Pc @ glob[l].P)
    Pc = Pc.tocsr()
    PcT = Pc.T.tocsr()
print(f"setup {time.time() - t0:.1f}s levels(local0)={len(locs[0][2])} cuts={cuts}", flush=True)


def local_apply(r):
    z = np.empty_like(r)
    for f, e, lv in locs:
        z[f:e] = mr.cycle(lv, 0, r[f:e])
    return z


def precond(r):
    if a.P == 1:
        return mr.cycle(glob, 0, r)
    z = local_apply(r)
    if a.mode == "add":
        z = z + Pc @ mr.cycle(glob, a.lc, PcT @ r)
    elif a.mode == "mult":
        z = z + Pc @ mr.cycle(glob, a.lc, PcT @ (r - A @ z))
    elif a.mode == "mult2":       # coarse first, then local on the updated residual
        z0 = Pc @ mr.cycle(glob, a.lc, PcT @ r)
        z = z0 + local_apply(r - A @ z0)
    elif a.mode == "sym":         # local, coarse, local
        z = z + Pc @ mr.cycle(glob, a.lc, PcT @ (r - A @ z))
        z = z + local_apply(r - A @ z)
    return z


def fgmres(b, rtol, m, maxit):
    x = np.zeros_like(b)
    bn = np.linalg.norm(b)
    its = 0
    while its < maxit:
        r = b - A @ x
        beta = np.linalg.norm(r)
        print(f"  its {its} relres {beta / bn:.3e}", flush=True)
        if beta <= rtol * bn:
            break
        V = np.zeros((m + 1, n))
        Z = np.zeros((m, n))
        H = np.zeros((m + 1, m))
        V[0] = r / beta
        g = np.zeros(m + 1)
        g[0] = beta
        cs_, sn_ = np.zeros(m), np.zeros(m)
        k = 0
        for j in range(m):
            Z[j] = precond(V[j])
            w = A @ Z[j]
            for _ in range(2):
                h = V[:j + 1] @ w
                w -= h @ V[:j + 1]
                H[:j + 1, j] += h
            H[j + 1, j] = np.linalg.norm(w)
            V[j + 1] = w / H[j + 1, j]
            for i in range(j):
                t = cs_[i] * H[i, j] + sn_[i] * H[i + 1, j]
                H[i + 1, j] = -sn_[i] * H[i, j] + cs_[i] * H[i + 1, j]
                H[i, j] = t
            d = np.hypot(H[j, j], H[j + 1, j])
            cs_[j], sn_[j] = H[j, j] / d, H[j + 1, j] / d
            H[j, j], H[j + 1, j] = d, 0.0
            g[j + 1] = -sn_[j] * g[j]
            g[j] = cs_[j] * g[j]
            its += 1
            k = j + 1
            if abs(g[j + 1]) <= rtol * bn or its >= maxit:
                break
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x += y @ Z[:k]
    return x, its


b = np.random.default_rng(1).standard_normal(n)
t0 = time.time()
x, its = fgmres(b, a.rtol, a.restart, a.maxit)
print(f"RESULT grid={a.grid} P={a.P} mode={a.mode} lc={a.lc} its={its} relres={np.linalg.norm(b - A @ x) / np.linalg.norm(b):.2e} time={time.time() - t0:.0f}s")
