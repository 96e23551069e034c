#!/usr/bin/env python3
"""CPU prototype (scipy) of multilevel-cycle variants, iteration counts only (round 2).

  --rescale W     after each Galerkin product, per inter-column edge (i, j): s = (a_ij + a_ji) / 2,
                  n = (a_ij - a_ji) / 2, s' = max(s / w, |n|) with w the aggregate width (2 or 4); the removed
                  diffusion goes back onto the two diagonals.  W = 0 off, 1 on, other = fixed divisor.
  --kcycle L      Krylov (2 inner FGMRES steps) acceleration of the coarse correction on levels < L
  --refine F      cell-level coefficients of a grid F times finer (u x F, ah x F^2) on the same cell count
  --donor         A = its own twin
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
ap.add_argument("--adv", default="upwind3")
ap.add_argument("--hmix", default="isop")
ap.add_argument("--rescale", type=float, default=0)
ap.add_argument("--kcycle", type=int, default=0)
ap.add_argument("--refine", type=float, default=1.0)
ap.add_argument("--nu", type=int, default=3)
ap.add_argument("--rtol", type=float, default=1e-8)
ap.add_argument("--restart", type=int, default=100)
ap.add_argument("--maxit", type=int, default=600)
ap.add_argument("--big-from", type=int, default=3)
ap.add_argument("--coarsest", type=int, default=1500)
ap.add_argument("--solve-twin", action="store_true", help="solve L x = b instead of A x = b")
ap.add_argument("--omega", type=float, default=1.0)
ap.add_argument("--agg", default="geo", help="geo | pair (flow-aligned pairwise matching of columns on the symmetric part)")
ap.add_argument("--passes", type=int, default=2, help="pairwise passes per level (aggregates of 2^passes columns)")
ap.add_argument("--beta", type=float, default=0.0, help="a pair needs weight >= beta * strongest weight of the column")
ap.add_argument("--theta", type=float, default=0.0, help="geosplit: lateral coupling counts when >= theta * strongest lateral coupling of the row")
ap.add_argument("--no-absorb", action="store_true")
ap.add_argument("--tau", type=float, default=0.01, help="geosplit: a stub is absorbed when no outside row feels it more than tau x its diagonal")
ap.add_argument("--pocket", type=int, default=0, help="geosplit: same-depth connected sets of at most this many cells become one coarse cell across groups")
ap.add_argument("--inner", type=int, default=0, help="preconditioner = this many FGMRES steps on the twin L, each preconditioned by one cycle")
ap.add_argument("--k33", action="store_true")
ap.add_argument("--post-a", type=int, default=0, help="damped column-block Jacobi sweeps on A itself after the cycle")
ap.add_argument("--post-omega", type=float, default=0.7)
ap.add_argument("--sub", action="store_true", help="GMRES iteration counts of every sub-hierarchy and two-grid pair")
ap.add_argument("--diagnose", type=int, default=0, help="power-iterate I - M L this many steps and describe the slow mode")
a = ap.parse_args()
imt, jmt, km = (int(t) for t in a.grid.split("x"))
t0 = time.time()
p = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, u_scale=3.0 * a.refine, ah=4.0e6 * a.refine ** 2, isop_k33=a.k33)
A = p.scipy_csr()
n = A.shape[0]
cs = p.col_start()
colid0 = np.repeat(np.arange(len(cs) - 1), np.diff(cs))
print(f"n={n} nnz={A.nnz} gen {time.time() - t0:.1f}s", flush=True)


def rescale(L, colid, w):
    """non-Galerkin correction of the inter-column couplings (see module docstring)"""
    C = L.tocoo()
    inter = colid[C.row] != colid[C.col]
    X = sp.csr_matrix((C.data[inter], (C.row[inter], C.col[inter])), shape=L.shape)
    XT = X.T.tocsr()
    U = X + XT                    # 2 s
    D = X - XT                    # 2 n
    S2 = (U * (1.0 / (2.0 * w))).maximum(abs(D) * 0.5)
    Xn = S2 + D * 0.5
    dl = np.asarray((X - Xn).sum(1)).ravel()
    return (L - X + Xn + sp.diags(dl)).tocsr()


def col_graph(L, colid, ncol):
    C = L.tocoo()
    m = colid[C.row] != colid[C.col]
    W = sp.csr_matrix((abs(C.data[m]), (colid[C.row[m]], colid[C.col[m]])), shape=(ncol, ncol))
    W.sum_duplicates()
    return W


def pairwise(S):
    """greedy matching on a symmetric weighted graph; returns group id per node"""
    nn = S.shape[0]
    grp = np.full(nn, -1, np.int64)
    ptr, idx, dat = S.indptr, S.indices, S.data
    ng = 0
    for c in range(nn):
        if grp[c] >= 0:
            continue
        best, bw, mx = -1, 0.0, 0.0
        for q in range(ptr[c], ptr[c + 1]):
            j = idx[q]
            if j == c:
                continue
            mx = max(mx, dat[q])
            if grp[j] < 0 and dat[q] > bw:
                bw, best = dat[q], j
        grp[c] = ng
        if best >= 0 and bw >= a.beta * mx:
            grp[best] = ng
        ng += 1
    return grp, ng


def pair_aggregate(L, colid, ncol, passes):
    W = col_graph(L, colid, ncol)
    S = ((W + W.T) * 0.5).tocsr()
    agg = np.arange(ncol)
    for _ in range(passes):
        grp, ng = pairwise(S)
        agg = grp[agg]
        G = sp.csr_matrix((np.ones(S.shape[0]), (np.arange(S.shape[0]), grp)), shape=(S.shape[0], ng))
        S = (G.T @ S @ G).tocsr()
        S.setdiag(0)
        S.eliminate_zeros()
    return agg


def greedy_colour(L, colid, ncol):
    W = col_graph(L, colid, ncol)
    S = (W + W.T).tocsr()
    colour = np.full(ncol, -1, np.int64)
    for c in range(ncol):
        used = set(colour[S.indices[S.indptr[c]:S.indptr[c + 1]]].tolist())
        k = 0
        while k in used:
            k += 1
        colour[c] = k
    return colour


def geosplit(L, ci, cj, ck, colid, sh, theta, absorb=True):
    """geometric (2^sh x 2^sh) groups of columns, but the members of a group that are wet at depth k form one coarse cell
    per CONNECTED set (strong lateral couplings inside the group); sets are threaded through depth into coarse columns,
    the largest overlap continues a column, the rest start stub columns.  Dangling stubs of this level (only coupled to
    one other column) are absorbed into the cell above them.  Returns per fine row the coarse key arrays."""
    from scipy.sparse.csgraph import connected_components
    n = L.shape[0]
    I, J = ci >> sh, cj >> sh
    grp = J.astype(np.int64) * (int(I.max()) + 2) + I
    C = L.tocoo()
    inter = colid[C.row] != colid[C.col]
    rowmax = np.zeros(n)
    np.maximum.at(rowmax, C.row[inter], abs(C.data[inter]))
    # stubs of this level that can be absorbed into the cell they hang from: every coupling FROM another column TO the
    # stub is weak against that column's own diagonal (the stub is a leaf: it follows its neighbours, they do not feel it)
    ncol = int(colid.max()) + 1
    ktop = np.full(ncol, 1 << 30, np.int64)
    np.minimum.at(ktop, colid, ck)
    diag = abs(L.diagonal())
    ir, ic, iv = C.row[inter], C.col[inter], abs(C.data[inter])
    felt = np.zeros(ncol)                                   # how strongly any outside row feels the column
    np.maximum.at(felt, colid[ic], iv / diag[ir])
    skey = colid.astype(np.int64) * 4096 + ck
    order = np.argsort(skey, kind="stable")
    skeys = skey[order]

    def row_of(col, k):
        q = col.astype(np.int64) * 4096 + k
        pos = np.searchsorted(skeys, q)
        pos = np.minimum(pos, n - 1)
        ok = skeys[pos] == q
        return np.where(ok, order[pos], -1)

    dang = np.zeros(ncol, bool)
    anchor_row = np.full(ncol, -1, np.int64)
    if absorb:
        cand = (ktop > 0) & (felt < a.tau)
        # anchor: target of the strongest inter-column entry of the stub
        best = np.zeros(ncol)
        o = np.argsort(iv, kind="stable")
        tgt = np.full(ncol, -1, np.int64)
        tgt[colid[ir[o]]] = ic[o]                            # last write = strongest
        cand &= tgt >= 0
        dang[cand] = True
        anchor_row[cand] = tgt[cand]
        # an anchor must not be a stub that is itself absorbed
        bad = dang & dang[colid[np.maximum(anchor_row, 0)]]
        dang[bad] = False
    base = inter & (abs(ck[C.row] - ck[C.col]) <= 1) & ~dang[colid[C.row]] & ~dang[colid[C.col]]
    rb, cb, vb = C.row[base], C.col[base], abs(C.data[base])
    trb = row_of(colid[cb], ck[rb])
    okb = (vb >= theta * rowmax[rb]) & (trb >= 0)
    sameg = grp[rb] == grp[cb]
    if a.pocket > 0:
        G0 = sp.csr_matrix((np.ones(okb.sum()), (rb[okb], trb[okb])), shape=(n, n))
        _, comp0 = connected_components(G0, directed=False)
        size0 = np.bincount(comp0)
        small = size0[comp0] <= a.pocket
        ok = okb & (sameg | small[rb])
    else:
        ok = okb & sameg
    G = sp.csr_matrix((np.ones(ok.sum()), (rb[ok], trb[ok])), shape=(n, n))
    ncomp, comp = connected_components(G, directed=False)
    # thread components through depth
    below = np.full(n, -1, np.int64)
    nxt = np.arange(n - 1)
    same = (colid[nxt] == colid[nxt + 1]) & (ck[nxt + 1] == ck[nxt] + 1)
    below[nxt[same]] = nxt[same] + 1
    has = below >= 0
    pc = np.unique(comp[has].astype(np.int64) * ncomp + comp[below[has]], return_counts=True)
    par, chi, cnt = pc[0] // ncomp, pc[0] % ncomp, pc[1]
    # best parent of each child, best child of each parent (largest overlap, ties -> lowest id)
    o = np.lexsort((par, -cnt, chi))
    first = np.ones(o.size, bool); first[1:] = chi[o][1:] != chi[o][:-1]
    bestpar = np.full(ncomp, -1, np.int64); bestpar[chi[o][first]] = par[o][first]
    o = np.lexsort((chi, -cnt, par))
    first = np.ones(o.size, bool); first[1:] = par[o][1:] != par[o][:-1]
    bestchi = np.full(ncomp, -1, np.int64); bestchi[par[o][first]] = chi[o][first]
    kcomp = np.zeros(ncomp, np.int64); kcomp[comp] = ck
    gcomp = np.zeros(ncomp, np.int64); gcomp[comp] = grp
    ccol = np.full(ncomp, -1, np.int64)
    ncc = 0
    byk = np.argsort(kcomp, kind="stable")
    ks = kcomp[byk]
    for d in range(int(ks.max()) + 1):
        ids = byk[np.searchsorted(ks, d):np.searchsorted(ks, d + 1)]
        cont = (bestpar[ids] >= 0) & (bestchi[np.maximum(bestpar[ids], 0)] == ids)
        ccol[ids[cont]] = ccol[bestpar[ids[cont]]]
        new = ids[~cont]
        ccol[new] = ncc + np.arange(new.size)
        ncc += new.size
    crow_key = ccol[comp] * 4096 + ck
    # absorbed stubs take the coarse cell of their anchor
    drows = np.flatnonzero(dang[colid])
    crow_key[drows] = crow_key[anchor_row[colid[drows]]]
    npocket = 0
    uk, inv = np.unique(crow_key, return_inverse=True)
    colraw = uk // 4096
    ucol, colid2 = np.unique(colraw, return_inverse=True)
    ck2 = uk % 4096
    # group coordinates of each coarse column
    gI = np.zeros(ncc, np.int64); gJ = np.zeros(ncc, np.int64)
    keep = ~dang[colid]
    gI[ccol[comp[keep]]] = I[keep]; gJ[ccol[comp[keep]]] = J[keep]
    ci2, cj2 = gI[colraw], gJ[colraw]
    print(f"      split: {ncol} columns ({dang.sum()} dangling stubs absorbed) -> {ucol.size} coarse columns in {np.unique(grp).size} groups", flush=True)
    return uk.size, inv, ci2, cj2, ck2, colid2


def build(A, ci, cj, ck, colid):
    levels = []
    L = mr.low_order(A, colid)
    lvl = 0
    while True:
        lv = mr.Level()
        lv.A = L.tocsr()
        lv.n = L.shape[0]
        ncol = int(colid.max()) + 1
        colour = (ci + cj) % 2 if a.agg.startswith("geo") else greedy_colour(lv.A, colid, ncol)
        ncolours = int(colour.max()) + 1
        colour = colour[colid] if not a.agg.startswith("geo") else colour
        C = lv.A.tocoo()
        same = colid[C.row] == colid[C.col]
        Bd = sp.csr_matrix((C.data[same], (C.row[same], C.col[same])), shape=L.shape)
        lv.colours = []
        for c in range(ncolours):
            rows = np.flatnonzero(colour == c)
            lv.colours.append((rows, spla.splu(Bd[rows][:, rows].tocsc(), permc_spec="NATURAL") if rows.size else None, lv.A[rows]))
        lv.colid, lv.ci, lv.cj, lv.ck = colid, ci, cj, ck
        levels.append(lv)
        offd = lv.A - sp.diags(lv.A.diagonal())
        print(f"  level {lvl}: n={lv.n} nnz={lv.A.nnz} ncol={ncol} colours={ncolours} min offdiag {offd.data.min() if offd.nnz else 0:.2e}", flush=True)
        if len(levels) >= 12 or lv.n <= a.coarsest or ncol <= 4:
            break
        sh = 2 if (a.big_from >= 0 and lvl >= a.big_from) else 1
        if a.agg == "geosplit":
            nc_, inv, ci2, cj2, ck2, colid2 = geosplit(lv.A, ci, cj, ck, colid, sh, a.theta, absorb=not a.no_absorb)
            if nc_ >= lv.n:
                break
            uk = np.arange(nc_)
        elif a.agg == "geo":
            I, J = ci >> sh, cj >> sh
            key = (J.astype(np.int64) * (int(I.max()) + 2) + I) * 4096 + ck
            uk, inv = np.unique(key, return_inverse=True)
            if uk.size >= lv.n:
                break
            ck2 = uk % 4096
            rest = uk // 4096
            ci2, cj2 = rest % (int(I.max()) + 2), rest // (int(I.max()) + 2)
            _, colid2 = np.unique(rest, return_inverse=True)
        else:
            agg = pair_aggregate(lv.A, colid, ncol, a.passes)
            key = agg[colid].astype(np.int64) * 4096 + ck
            uk, inv = np.unique(key, return_inverse=True)
            if uk.size >= lv.n:
                break
            ck2 = uk % 4096
            colid2 = (uk // 4096).astype(np.int64)
            # representative position of an aggregate: mean of member positions (only used for reporting)
            ci2 = cj2 = np.zeros(uk.size, np.int64)
        lv.P = sp.csr_matrix((np.ones(lv.n), (np.arange(lv.n), inv)), shape=(lv.n, uk.size))
        lv.PT = lv.P.T.tocsr()
        L = (lv.PT @ lv.A @ lv.P).tocsr()
        ci, cj, ck, colid = ci2, cj2, ck2, colid2
        if a.rescale:
            L = rescale(L, colid, float(1 << sh) if a.rescale == 1 else a.rescale)
        lvl += 1
    levels[-1].lu = spla.splu(levels[-1].A.tocsc())
    return levels


def sweep(lv, x, b, reverse):
    for rows, lu, Arows in (lv.colours[::-1] if reverse else lv.colours):
        if rows.size:
            x[rows] += lu.solve(b[rows] - Arows @ x)
    return x


def coarse_solve(levels, l, r):
    """approximate solve on level l (l >= 1)"""
    if l == len(levels) - 1 or l >= a.kcycle:
        return cycle(levels, l, r)
    # K-cycle: two steps of flexible GCR on level l preconditioned by the cycle
    lv = levels[l]
    c1 = cycle(levels, l, r)
    v1 = lv.A @ c1
    al1 = (v1 @ r) / (v1 @ v1)
    r1 = r - al1 * v1
    if np.linalg.norm(r1) <= 0.25 * np.linalg.norm(r):
        return al1 * c1
    c2 = cycle(levels, l, r1)
    v2 = lv.A @ c2
    g = (v2 @ v1) / (v1 @ v1)
    v2o = v2 - g * v1
    c2o = c2 - g * c1
    al2 = (v2o @ r1) / max(v2o @ v2o, 1e-300)
    return al1 * c1 + al2 * c2o


def cycle(levels, l, b):
    lv = levels[l]
    if l == len(levels) - 1:
        return lv.lu.solve(b)
    x = np.zeros_like(b)
    for _ in range(a.nu):
        x = sweep(lv, x, b, False)
    r = b - lv.A @ x
    x = x + a.omega * (lv.P @ coarse_solve(levels, l + 1, lv.PT @ r))
    for _ in range(a.nu):
        x = sweep(lv, x, b, True)
    return x


t0 = time.time()
levels = build(A, p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64), colid0)
print(f"setup {time.time() - t0:.1f}s levels={len(levels)}", flush=True)
if a.solve_twin:
    A = levels[0].A
if os.environ.get("PROTO_DUMP"):
    import pickle
    pickle.dump([dict(A=lv.A, P=getattr(lv, "P", None), rows=[c[0] for c in lv.colours], colid=lv.colid, ci=lv.ci, cj=lv.cj, ck=lv.ck) for lv in levels], open(os.environ["PROTO_DUMP"], "wb"))
if a.diagnose < 0:
    for l0 in range(len(levels) - 2, -1, -1):
        Lf = levels[l0].A
        e = np.random.default_rng(3).standard_normal(Lf.shape[0])
        for it in range(-a.diagnose):
            e /= np.linalg.norm(e)
            e2 = e - cycle(levels, l0, Lf @ e)
            rho = e2 @ e
            e = e2
        print(f"  sub-hierarchy from level {l0}: factor {np.linalg.norm(e2):.4f} (signed {rho:.4f})", flush=True)
    sys.exit(0)
if a.diagnose:
    Lf = levels[0].A
    e = np.random.default_rng(3).standard_normal(n)
    for it in range(a.diagnose):
        e /= np.linalg.norm(e)
        e2 = e - cycle(levels, 0, Lf @ e)
        rho = np.linalg.norm(e2)
        print(f"  power it {it} factor {rho:.4f}", flush=True)
        e = e2
    e /= np.linalg.norm(e)
    k_ = p.ind_k; j_ = p.ind_j; i_ = p.ind_i
    ek = np.bincount(k_, weights=e * e, minlength=km)
    print("energy by level k:", np.array2string(ek, precision=3, max_line_width=200))
    ej = np.bincount(j_, weights=e * e, minlength=jmt)
    print("energy by j:", np.array2string(ej, precision=3, max_line_width=200))
    lv = levels[0]
    cnt = np.asarray(lv.PT.sum(1)).ravel()
    ebar = lv.P @ ((lv.PT @ e) / cnt)
    print("fraction of e outside the coarse space:", np.linalg.norm(e - ebar))
    print("|L e| / |e| =", np.linalg.norm(Lf @ e), " |L| row-abs max", abs(Lf).sum(1).max())
    np.save("/tmp/p/slowmode.npy", e)
    sys.exit(0)


def fgmres(Aop, prec, b, rtol, m, maxit, verbose=True):
    n = b.size
    bn = np.linalg.norm(b)
    x = np.zeros(n)
    its = 0
    t0 = time.time()
    while its < maxit:
        r = b - Aop @ x
        beta = np.linalg.norm(r)
        if verbose:
            print(f"  its {its} relres {beta / bn:.3e}  ({time.time() - t0:.0f}s)", flush=True)
        if beta <= rtol * bn or not np.isfinite(beta):
            break
        V = np.zeros((m + 1, n)); Z = np.zeros((m, n)); H = np.zeros((m + 1, m))
        V[0] = r / beta
        g = np.zeros(m + 1); g[0] = beta
        cs_, sn_ = np.zeros(m), np.zeros(m)
        k = 0
        for j in range(m):
            Z[j] = prec(V[j])
            w = Aop @ Z[j]
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
            if verbose and its % 10 == 0:
                print(f"    it {its} est {abs(g[j + 1]) / bn:.3e}", flush=True)
            if abs(g[j + 1]) <= rtol * bn or its >= maxit:
                break
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x += y @ Z[:k]
    return x, its, np.linalg.norm(b - Aop @ x) / bn


if a.sub:
    for l0 in range(len(levels) - 2, -1, -1):
        Lf = levels[l0].A
        bb = np.random.default_rng(1).standard_normal(Lf.shape[0])
        _, itv, rv = fgmres(Lf, lambda r: cycle(levels, l0, r), bb, a.rtol, a.restart, a.maxit, verbose=False)
        # two-grid with an exact solve of level l0 + 1
        lu = spla.splu(levels[l0 + 1].A.tocsc()) if levels[l0 + 1].n < 150000 else None
        if lu is not None:
            save = levels[l0 + 1:]
            tg = levels[:l0 + 1] + [levels[l0 + 1]]
            lvx = mr.Level(); lvx.lu = lu; lvx.A = levels[l0 + 1].A; lvx.n = levels[l0 + 1].n
            tg[-1] = lvx
            _, it2, r2 = fgmres(Lf, lambda r: cycle(tg, l0, r), bb, a.rtol, a.restart, a.maxit, verbose=False)
        else:
            it2, r2 = -1, 0
        print(f"  from level {l0} (n={Lf.shape[0]}): multilevel its {itv} ({rv:.1e}); two-grid exact-coarse its {it2} ({r2:.1e})", flush=True)
    sys.exit(0)
b = np.random.default_rng(1).standard_normal(n)
t0 = time.time()
if a.post_a > 0:
    CA = A.tocoo()
    sameA = colid0[CA.row] == colid0[CA.col]
    BA = spla.splu(sp.csr_matrix((CA.data[sameA], (CA.row[sameA], CA.col[sameA])), shape=A.shape).tocsc(), permc_spec="NATURAL")

    def prec_post(r):
        z = cycle(levels, 0, r)
        for _ in range(a.post_a):
            z = z + a.post_omega * BA.solve(r - A @ z)
        return z
if a.inner > 0:
    Ltwin = levels[0].A
    prec = lambda r: fgmres(Ltwin, lambda q: cycle(levels, 0, q), r, 1e-30, a.inner, a.inner, verbose=False)[0]
elif a.post_a > 0:
    prec = prec_post
else:
    prec = lambda r: cycle(levels, 0, r)
x, its, rr = fgmres(A, prec, b, a.rtol, a.restart, a.maxit)
print(f"RESULT grid={a.grid} adv={a.adv} refine={a.refine} rescale={a.rescale} kcycle={a.kcycle} nu={a.nu} omega={a.omega} twin={a.solve_twin} "
      f"its={its} relres={rr:.2e} time={time.time() - t0:.0f}s")
