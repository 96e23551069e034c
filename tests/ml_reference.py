"""Independent scipy restatement of the multilevel water-column preconditioner, for cross-checking the HIP
implementation (csrc/multilevel.hip) on small grids.  TEST INFRASTRUCTURE: slow, sequential, never shipped.

Same rules, written from the description in DESIGN.md section 2, not from the C++:
  * low-order twin: L = A + D - diag(rowsum D), D_ij = max(0, -a_ij, -a_ji) for i, j in different water columns
  * aggregates: 2 x 2 columns in (i, j) per tracer (4 x 4 from level `big_from` down), levels k kept,
    piecewise-constant P, Galerkin L_c = P^T L P; stop when a level has <= coarsest_rows rows or <= 4 columns
  * smoother: 2-colour ((i + j) % 2) column-block Gauss-Seidel, nu sweeps before (colour 0 first) and after
    (colour 1 first) the coarse correction; exact dense solve on the last level
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
    L = A + D - sp.diags(np.asarray(D.sum(1)).ravel())
    return L.tocsr()


class Level:
    pass


def build(A, ci, cj, ck, colid, nu=3, coarsest_rows=1500, big_from=3, max_levels=12):
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
        levels.append(lv)
        if len(levels) >= max_levels or lv.n <= coarsest_rows or ncol <= 4:
            break
        sh = 2 if (big_from >= 0 and lvl >= big_from) else 1
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


def _sweep(lv, x, b, reverse):
    order = lv.colours[::-1] if reverse else lv.colours
    for rows, lu, Arows in order:
        if rows.size:
            x[rows] += lu.solve(b[rows] - Arows @ x)
    return x


def cycle(levels, l, b, nu=3):
    lv = levels[l]
    if l == len(levels) - 1:
        return lv.dense_inv @ b
    x = np.zeros_like(b)
    for _ in range(nu):
        x = _sweep(lv, x, b, False)
    r = b - lv.A @ x
    x = x + lv.P @ cycle(levels, l + 1, lv.P.T @ r, nu)
    for _ in range(nu):
        x = _sweep(lv, x, b, True)
    return x
