"""Independent scipy restatement of the multilevel water-column preconditioner, for cross-checking the HIP
implementation (csrc/multilevel.hip) on small grids.  TEST INFRASTRUCTURE: slow, sequential, never shipped.

Same rules, written from the description in DESIGN.md section 2, not from the C++:
  * low-order twin: L = A + D - diag(rowsum D), D_ij = max(0, -a_ij, -a_ji) for i, j in different water columns
  * aggregates: groups of 2 x 2 columns in (i, j) per tracer (4 x 4 from level `big_from` down), levels k kept; inside a
    group the cells of one depth form one coarse cell per laterally CONNECTED set (`split_groups` below: connected
    sets, same-depth sets of <= `pocket` cells merged across groups, sets threaded through depth into coarse columns
    by largest overlap, leaf stubs absorbed into the cell they hang from); piecewise-constant P, Galerkin
    L_c = P^T L P; stop when a level has <= coarsest_rows rows or <= 4 columns
  * smoother: 2-colour ((i + j) % 2) column-block Gauss-Seidel, nu sweeps before (colour 0 first) and after
    (colour 1 first) the coarse correction, which is weighted by omega = 1.1; exact dense solve on the last level
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def low_order(A, colid):
    A = A.tocsr()
    off = (A - sp.diags(A.diagonal())).tocoo()
    m = colid[off.row] != colid[off.col]
    inter = sp.csr_matrix((off.data[m], (off.row[m], off.col[m])), shape=A.shape)
    neg = (-inter).maximum(0)
    D = neg.maximum(neg.T)
    # the C++ only touches entries that exist in A's pattern
    pat = inter.copy()
    pat.data[:] = 1.0
    D = D.multiply(pat)
    L = (A + D - sp.diags(np.asarray(D.sum(1)).ravel())).tocsr()
    L.eliminate_zeros()               # a removed coupling is not stored (it would count as a connection)
    return L


class Level:
    pass


def split_groups(L, I, J, ck, colid, pocket=4, theta=0.0, tau=0.01):
    """Coarse cells of one coarsening step.  I, J: group coordinates of every row; ck: depth of every row; colid: column
    of every row (rows of a column are contiguous and ordered by depth).  Returns (coarse row of every fine row, and per
    coarse row: I, J, depth, coarse column)."""
    from scipy.sparse.csgraph import connected_components
    n = L.shape[0]
    grp = J.astype(np.int64) * (int(I.max()) + 2) + I
    C = L.tocoo()
    inter = colid[C.row] != colid[C.col]
    ir, ic, iv = C.row[inter], C.col[inter], abs(C.data[inter])
    rowmax = np.zeros(n)
    np.maximum.at(rowmax, ir, iv)
    ncol = int(colid.max()) + 1
    ktop = np.full(ncol, 1 << 30, np.int64)
    np.minimum.at(ktop, colid, ck)
    first_row = np.full(ncol, n, np.int64)
    np.minimum.at(first_row, colid, np.arange(n))
    last_row = np.zeros(ncol, np.int64)
    np.maximum.at(last_row, colid, np.arange(n))

    def row_of(col, k):
        r = first_row[col] + (k - ktop[col])
        return np.where((k >= ktop[col]) & (r <= last_row[col]), r, -1)

    # leaf stubs: columns that start below the surface and that no outside row feels
    diag = abs(L.diagonal())
    felt = np.zeros(ncol)
    np.maximum.at(felt, colid[ic], iv / diag[ir])
    anchor = np.full(ncol, -1, np.int64)
    o = np.argsort(iv, kind="stable")
    anchor[colid[ir[o]]] = ic[o]                       # strongest coupling of the column, ties -> later entry
    leaf = (ktop > 0) & (felt < tau) & (anchor >= 0)
    leaf &= ~(leaf & leaf[colid[np.maximum(anchor, 0)]])
    # lateral edges between cells of equal depth
    m = (abs(ck[ir] - ck[ic]) <= 1) & ~leaf[colid[ir]] & ~leaf[colid[ic]] & (iv >= theta * rowmax[ir])
    r, c = ir[m], ic[m]
    t = row_of(colid[c], ck[r])
    ok = t >= 0
    r, t = r[ok], t[ok]
    same = grp[r] == grp[t]
    if pocket > 0:
        _, c0 = connected_components(sp.csr_matrix((np.ones(r.size), (r, t)), shape=(n, n)), directed=False)
        small = np.bincount(c0)[c0] <= pocket
        keep = same | small[r]
    else:
        keep = same
    _, comp = connected_components(sp.csr_matrix((np.ones(keep.sum()), (r[keep], t[keep])), shape=(n, n)), directed=False)
    _, firsts, comp = np.unique(comp, return_index=True, return_inverse=True)
    comp = np.argsort(np.argsort(firsts))[comp]        # sets numbered by their lowest row
    ncomp = int(comp.max()) + 1
    # overlaps between a set and the sets directly below it
    nxt = np.arange(n - 1)
    has = colid[nxt] == colid[nxt + 1]
    pairs, cnt = np.unique(comp[nxt[has]].astype(np.int64) * ncomp + comp[nxt[has] + 1], return_counts=True)
    par, chi = pairs // ncomp, pairs % ncomp
    o = np.lexsort((par, -cnt, chi))
    f = np.ones(o.size, bool); f[1:] = chi[o][1:] != chi[o][:-1]
    bestpar = np.full(ncomp, -1, np.int64); bestpar[chi[o][f]] = par[o][f]
    o = np.lexsort((chi, -cnt, par))
    f = np.ones(o.size, bool); f[1:] = par[o][1:] != par[o][:-1]
    bestchi = np.full(ncomp, -1, np.int64); bestchi[par[o][f]] = chi[o][f]
    kcomp = np.zeros(ncomp, np.int64); kcomp[comp] = ck
    ccol = np.full(ncomp, -1, np.int64)
    ncc = 0
    for cid in np.argsort(kcomp, kind="stable"):
        pr = bestpar[cid]
        if pr >= 0 and bestchi[pr] == cid:
            ccol[cid] = ccol[pr]
        else:
            ccol[cid] = ncc
            ncc += 1
    key = ccol[comp] * 4096 + ck
    lrows = np.flatnonzero(leaf[colid])
    key[lrows] = key[anchor[colid[lrows]]]
    uk, inv = np.unique(key, return_inverse=True)
    # a coarse column sits at the (I, J) of its lowest fine row (a merged pocket can span groups)
    gI = np.zeros(ncc, np.int64); gJ = np.zeros(ncc, np.int64)
    keepr = np.flatnonzero(~leaf[colid])[::-1]
    gI[ccol[comp[keepr]]] = I[keepr]; gJ[ccol[comp[keepr]]] = J[keepr]
    colraw = uk // 4096
    _, colid2 = np.unique(colraw, return_inverse=True)
    return inv, gI[colraw], gJ[colraw], uk % 4096, colid2


def build(A, ci, cj, ck, colid, nu=3, coarsest_rows=8000, big_from=3, max_levels=12, split=True):
    levels = []
    L = low_order(A, colid)
    lvl = 0
    while True:
        lv = Level()
        lv.A = L.tocsr()
        lv.n = L.shape[0]
        ncol = int(colid.max()) + 1
        colour = (ci + cj) % 2
        C = lv.A.tocoo()
        same = colid[C.row] == colid[C.col]
        Bd = sp.csr_matrix((C.data[same], (C.row[same], C.col[same])), shape=L.shape)
        lv.colours = []
        for c in range(2):
            rows = np.flatnonzero(colour == c)
            lv.colours.append((rows, spla.splu(Bd[rows][:, rows].tocsc()) if rows.size else None, lv.A[rows]))
        lv.ci, lv.cj, lv.ck, lv.colid = ci, cj, ck, colid
        levels.append(lv)
        if len(levels) >= max_levels or lv.n <= coarsest_rows or ncol <= 4:
            break
        sh = 2 if (big_from >= 0 and lvl >= big_from) else 1
        I, J = ci >> sh, cj >> sh
        if split:
            inv, ci2, cj2, ck2, colid2 = split_groups(lv.A, I, J, ck, colid)
            ncoarse = int(inv.max()) + 1
        else:
            key = (J.astype(np.int64) * (int(I.max()) + 2) + I) * 4096 + ck
            uk, inv = np.unique(key, return_inverse=True)
            ncoarse = uk.size
            ck2 = uk % 4096
            rest = uk // 4096
            ci2, cj2 = rest % (int(I.max()) + 2), rest // (int(I.max()) + 2)
            _, colid2 = np.unique(rest, return_inverse=True)
        if ncoarse >= lv.n:
            break
        lv.P = sp.csr_matrix((np.ones(lv.n), (np.arange(lv.n), inv)), shape=(lv.n, ncoarse))
        lv.cmap, lv.coarse_colid = inv, colid2
        L = (lv.P.T @ lv.A @ lv.P).tocsr()
        L.eliminate_zeros()
        ci, cj, ck, colid = ci2, cj2, ck2, colid2
        lvl += 1
    levels[-1].dense_inv = np.linalg.inv(levels[-1].A.toarray())
    return levels


def _sweep(lv, x, b, reverse):
    order = lv.colours[::-1] if reverse else lv.colours
    for rows, lu, Arows in order:
        if rows.size:
            x[rows] += lu.solve(b[rows] - Arows @ x)
    return x


def cycle(levels, l, b, nu=3, omega=1.1):
    lv = levels[l]
    if l == len(levels) - 1:
        return lv.dense_inv @ b
    x = np.zeros_like(b)
    for _ in range(nu):
        x = _sweep(lv, x, b, False)
    r = b - lv.A @ x
    x = x + omega * (lv.P @ cycle(levels, l + 1, lv.P.T @ r, nu, omega))
    for _ in range(nu):
        x = _sweep(lv, x, b, True)
    return x
