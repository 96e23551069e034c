"""CPU research prototype (not shipped, not a test): how many of the hierarchy's levels have to see across the rank cuts?

One GLOBAL hierarchy of the scipy restatement (tests/ml_reference.py) on a 3-degree matrix; for B latitude bands the operators
of levels l < k lose every coupling between rows of different bands (what per-rank levels do), levels l >= k keep them
(what levels replicated on all ranks would do).  k = number of levels = today's per-rank hierarchies without overlap,
k = 0 = the single-domain cycle.

  python tools/proto_replicated.py [--grid 100x116x60] [--bands 2,4] [--refine 4]
"""
import argparse, copy, os, sys, time
import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ml_reference as mlr                                                   # noqa: E402
from nk_ocn_tracer_jacobian_precond_amd import synth                        # noqa: E402
from proto_bands import fgmres                                               # noqa: E402


def cut_level(lv, band):
    """copy of a level whose operator has no entry between rows of different bands"""
    out = copy.copy(lv)
    C = lv.A.tocoo()
    keep = band[C.row] == band[C.col]
    out.A = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=lv.A.shape)
    out.colours = [(rows, lu, out.A[rows]) for rows, lu, _ in lv.colours]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="100x116x60")
    ap.add_argument("--bands", default="2,4")
    ap.add_argument("--refine", type=float, default=4.0)
    ap.add_argument("--maxit", type=int, default=400)
    a = ap.parse_args()
    imt, jmt, km = (int(t) for t in a.grid.split("x"))
    p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0, u_scale=3.0 * a.refine, ah=4.0e6 * a.refine ** 2)
    A = p.scipy_csr()
    n = p.flat_len
    colid = np.cumsum(p.ind_k == 0) - 1
    ci, cj, ck = p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64)
    levels = mlr.build(A, ci, cj, ck, colid)
    print("level rows", [lv.n for lv in levels], flush=True)
    b = np.random.default_rng(1).standard_normal(n)
    col_start = np.flatnonzero(p.ind_k == 0)
    x, its, rr = fgmres(A, b, lambda r: mlr.cycle(levels, 0, r.copy()), maxit=a.maxit)
    print(f"single domain: {its} iterations, relres {rr:.2e}", flush=True)
    for nb in (int(t) for t in a.bands.split(",")):
        cuts = [0] + [int(col_start[np.searchsorted(col_start, k * n // nb)]) for k in range(1, nb)] + [n]
        band = np.zeros(n, np.int64)
        for k in range(nb):
            band[cuts[k]:cuts[k + 1]] = k
        bands_l = [band]
        for lv in levels[:-1]:
            P = lv.P.tocsc()
            first = P.indices[P.indptr[:-1]]                   # a fine row of every coarse cell
            bands_l.append(bands_l[-1][first])
        for k in range(len(levels), -1, -1):
            mixed = [cut_level(lv, bands_l[l]) if l < k else lv for l, lv in enumerate(levels)]
            if k == len(levels):
                C = levels[-1].A.tocoo()
                keep = bands_l[-1][C.row] == bands_l[-1][C.col]
                D = sp.csr_matrix((C.data[keep], (C.row[keep], C.col[keep])), shape=C.shape).toarray()
                mixed[-1] = copy.copy(levels[-1]); mixed[-1].dense_inv = np.linalg.inv(D)
            t0 = time.perf_counter()
            x, its, rr = fgmres(A, b, lambda r: mlr.cycle(mixed, 0, r.copy()), maxit=a.maxit)
            print(f"bands {nb}: levels < {k} cut, levels >= {k} global: {its} iterations, relres {rr:.2e}, {time.perf_counter() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
