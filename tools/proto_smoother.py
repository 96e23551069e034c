#!/usr/bin/env python3
"""CPU prototype (scipy): smoother variants for the multilevel cycle, iteration counts only.
  --colours 2|4      checkerboard (i+j)%2 or the four (i%2, j%2) classes per level
  --smooth-a         level-0 Gauss-Seidel on A itself instead of its low-order twin
Not part of the product or the tests."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ml_reference as mr
from nk_ocn_tracer_jacobian_precond_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="100x116x60")
ap.add_argument("--colours", type=int, default=2)
ap.add_argument("--smooth-a", action="store_true")
ap.add_argument("--nu", type=int, default=3)
ap.add_argument("--rtol", type=float, default=1e-8)
ap.add_argument("--restart", type=int, default=200)
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0)
A = p.scipy_csr()
n = A.shape[0]
cs = p.col_start()
colid = np.repeat(np.arange(len(cs) - 1), np.diff(cs))


def build(A, ci, cj, ck, colid):
    levels = []
    L = mr.low_order(A, colid)
    lvl = 0
    while True:
        lv = mr.Level()
        lv.A = L.tocsr()
        lv.S = A.tocsr() if (lvl == 0 and a.smooth_a) else lv.A          # operator the smoother sees
        lv.n = L.shape[0]
        ncol = int(colid.max()) + 1
        colour = ((ci + cj) % 2) if a.colours == 2 else ((ci % 2) * 2 + (cj % 2))
        C = lv.S.tocoo()
        same = colid[C.row] == colid[C.col]
        Bd = sp.csr_matrix((C.data[same], (C.row[same], C.col[same])), shape=L.shape)
        lv.colours = []
        for c in range(a.colours):
            rows = np.flatnonzero(colour == c)
            lv.colours.append((rows, spla.splu(Bd[rows][:, rows].tocsc()) if rows.size else None, lv.S[rows]))
        levels.append(lv)
        if lv.n <= 1500 or ncol <= 4 or len(levels) >= 12:
            break
        sh = 2 if lvl >= 3 else 1
        I, J = ci >> sh, cj >> sh
        key = (J.astype(np.int64) * (int(I.max()) + 2) + I) * 4096 + ck
        uk, inv = np.unique(key, return_inverse=True)
        if uk.size >= lv.n:
            break
        lv.P = sp.csr_matrix((np.ones(lv.n), (np.arange(lv.n), inv)), shape=(lv.n, uk.size))
        L = (lv.P.T @ lv.A @ lv.P).tocsr()
        ck = uk % 4096
        rest = uk // 4096
        ci, cj = rest % (int(I.max()) + 2), rest // (int(I.max()) + 2)
        _, colid = np.unique(rest, return_inverse=True)
        lvl += 1
    levels[-1].dense_inv = np.linalg.inv(levels[-1].A.toarray())
    return levels


def sweep(lv, x, b, reverse):
    for rows, lu, Srows in (lv.colours[::-1] if reverse else lv.colours):
        if rows.size:
            x[rows] += lu.solve(b[rows] - Srows @ x)
    return x


def cycle(levels, l, b):
    lv = levels[l]
    if l == len(levels) - 1:
        return lv.dense_inv @ b
    x = np.zeros_like(b)
    for _ in range(a.nu):
        x = sweep(lv, x, b, False)
    r = b - lv.S @ x
    x = x + lv.P @ cycle(levels, l + 1, lv.P.T @ r)
    for _ in range(a.nu):
        x = sweep(lv, x, b, True)
    return x


t0 = time.time()
levels = build(A, p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64), colid)
print(f"setup {time.time() - t0:.1f}s levels={len(levels)}", flush=True)
b = np.random.default_rng(1).standard_normal(n)
bn = np.linalg.norm(b)
x = np.zeros(n)
its, m = 0, a.restart
while its < 3000:
    r = b - A @ x
    beta = np.linalg.norm(r)
    print(f"  its {its} relres {beta / bn:.3e}", flush=True)
    if beta <= a.rtol * bn or not np.isfinite(beta):
        break
    V = np.zeros((m + 1, n)); Z = np.zeros((m, n)); H = np.zeros((m + 1, m))
    V[0] = r / beta
    g = np.zeros(m + 1); g[0] = beta
    cs_, sn_ = np.zeros(m), np.zeros(m)
    k = 0
    for j in range(m):
        Z[j] = cycle(levels, 0, V[j])
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
        if abs(g[j + 1]) <= a.rtol * bn:
            break
    y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
    x += y @ Z[:k]
print(f"RESULT colours={a.colours} smooth_a={a.smooth_a} nu={a.nu} its={its} relres={np.linalg.norm(b - A @ x) / bn:.2e}")
