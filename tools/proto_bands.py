"""CPU research prototype (not shipped, not a test): what does a GLOBAL coarsest level buy a latitude-band partition?

The distributed flavour builds one multilevel hierarchy per rank from the rank's diagonal block (non-overlapping Schwarz).
Here, in the scipy restatement (tests/ml_reference.py), B bands of a 3-degree matrix are preconditioned
  (a) by their own cycles only                                      -- what nkp_create_dist does
  (b) with the B local coarsest solves replaced by ONE solve with the Galerkin operator P0^T T P0 of the global matrix T
      (P0 = the bands' prolongation chains side by side), i.e. the cross-band couplings restored on the last level only
and the FGMRES iteration counts are compared.

  python tools/proto_bands.py [--grid 100x116x60] [--bands 1,2,4] [--coarse twin|A]
"""
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
import ml_reference as mlr                                                   # noqa: E402
from nk_ocn_tracer_jacobian_precond_amd import synth                        # noqa: E402


def band_cycle(bands, r, glob=None, nu=3, omega=1.1, refresh=0):
    """One V-cycle per band; with `glob` = (inverse of the global coarsest operator, offsets) the coarsest solves of all
    bands are one solve."""
    z = np.zeros_like(r)
    state = []
    for bd in bands:
        lv_list, rows = bd["levels"], bd["rows"]
        b = r[rows]
        xs, bs = [], []
        for lv in lv_list[:-1]:
            x = np.zeros_like(b)
            for _ in range(nu):
                x = mlr._sweep(lv, x, b, False)
            xs.append(x)
            bs.append(b)
            b = lv.P.T @ (b - lv.A @ x)
        state.append((xs, bs, b))
    if glob is None:
        xc = [bd["levels"][-1].dense_inv @ st[2] for bd, st in zip(bands, state)]
    else:
        inv, off = glob
        full = inv @ np.concatenate([st[2] for st in state])
        xc = [full[off[k]:off[k + 1]] for k in range(len(bands))]
    if refresh:
        # variant: before (1) or before and between (2) the post-smoothing sweeps of the FINE level the overlap rows take
        # their owners' current values (one halo exchange of x each time)
        fine = []
        for bd, st, x in zip(bands, state, xc):
            lv_list = bd["levels"]
            xs, bs, _ = st
            for l in range(len(lv_list) - 2, 0, -1):
                lv = lv_list[l]
                x = xs[l] + omega * (lv.P @ x)
                for _ in range(nu):
                    x = mlr._sweep(lv, x, bs[l], True)
            lv = lv_list[0]
            fine.append(xs[0] + omega * (lv.P @ x))
        for sweep in range(nu):
            if sweep == 0 or refresh == 2:
                g = np.zeros_like(r)
                for bd, x in zip(bands, fine):
                    g[bd["rows"][bd["own"]]] = x[bd["own"]]
                fine = [g[bd["rows"]] for bd in bands]
            fine = [mlr._sweep(bd["levels"][0], x, st[1][0], True) for bd, st, x in zip(bands, state, fine)]
        for bd, x in zip(bands, fine):
            z[bd["rows"][bd["own"]]] = x[bd["own"]]
        return z
    for bd, st, x in zip(bands, state, xc):
        lv_list = bd["levels"]
        xs, bs, _ = st
        for l in range(len(lv_list) - 2, -1, -1):
            lv = lv_list[l]
            x = xs[l] + omega * (lv.P @ x)
            for _ in range(nu):
                x = mlr._sweep(lv, x, bs[l], True)
        if "own" in bd:
            z[bd["rows"][bd["own"]]] = x[bd["own"]]           # restricted additive Schwarz: keep the owned rows
        else:
            z[bd["rows"]] = x
    return z


def fgmres(A, b, M, rtol=1e-10, restart=200, maxit=2000):
    n = b.size
    x = np.zeros(n)
    its = 0
    bn = np.linalg.norm(b)
    while its < maxit:
        r = b - A @ x
        beta = np.linalg.norm(r)
        if beta <= rtol * bn:
            break
        V = [r / beta]
        Z = []
        H = np.zeros((restart + 1, restart))
        g = np.zeros(restart + 1)
        g[0] = beta
        k_used = 0
        for k in range(restart):
            z = M(V[k])
            w = A @ z
            for j in range(k + 1):
                H[j, k] = V[j] @ w
            for j in range(k + 1):
                w -= H[j, k] * V[j]
            H[k + 1, k] = np.linalg.norm(w)
            V.append(w / H[k + 1, k])
            Z.append(z)
            its += 1
            k_used = k + 1
            y, res, _, _ = np.linalg.lstsq(H[:k + 2, :k + 1], g[:k + 2], rcond=None)
            est = np.linalg.norm(g[:k + 2] - H[:k + 2, :k + 1] @ y)
            if est <= rtol * bn or its >= maxit:
                break
        for j in range(k_used):
            x += y[j] * Z[j]
    return x, its, float(np.linalg.norm(b - A @ x) / bn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="100x116x60")
    ap.add_argument("--bands", default="1,2,4")
    ap.add_argument("--coarse", default="twin", choices=["twin", "A"])
    ap.add_argument("--coarsest-rows", type=int, default=3000)
    ap.add_argument("--refine", type=float, default=1.0)
    ap.add_argument("--maxit", type=int, default=600)
    ap.add_argument("--refresh", type=int, default=0, help="overlap rows take their owners' values before (1) / before and between (2) the fine level's post-smoothing sweeps")
    ap.add_argument("--overlap", type=int, default=0, help="rings of neighbouring water columns added to every band (restricted additive Schwarz)")
    a = ap.parse_args()
    imt, jmt, km = (int(t) for t in a.grid.split("x"))
    p = synth.generate(imt=imt, jmt=jmt, km=km, adv="upwind3", hmix="isop", seed=0, u_scale=3.0 * a.refine, ah=4.0e6 * a.refine ** 2)
    A = p.scipy_csr()
    n = p.flat_len
    colid = np.cumsum(p.ind_k == 0) - 1
    ci, cj, ck = p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64)
    T = mlr.low_order(A, colid) if a.coarse == "twin" else A
    b = np.random.default_rng(1).standard_normal(n)
    col_start = np.flatnonzero(p.ind_k == 0)
    ncolumns = int(colid.max()) + 1
    C = A.tocoo()
    G = sp.csr_matrix((np.ones(C.nnz), (colid[C.row], colid[C.col])), shape=(ncolumns, ncolumns))
    G.data[:] = 1.0
    for nb in (int(t) for t in a.bands.split(",")):
        # contiguous row blocks of ~n / nb rows, snapped to water columns (the reference's n / P rule)
        cuts = [0]
        for k in range(1, nb):
            cuts.append(int(col_start[np.searchsorted(col_start, k * n // nb)]))
        cuts.append(n)
        bands = []
        t0 = time.perf_counter()
        for k in range(nb):
            rows = np.arange(cuts[k], cuts[k + 1])
            own = None
            if a.overlap > 0 and nb > 1:
                mark = np.zeros(ncolumns, bool)
                mark[colid[rows]] = True
                for _ in range(a.overlap):
                    mark = mark | (G @ mark.astype(np.float64) > 0)
                ext = np.flatnonzero(mark[colid])
                own = np.flatnonzero((ext >= cuts[k]) & (ext < cuts[k + 1]))
                rows = ext
            Ab = A[rows][:, rows].tocsr()
            _, cid = np.unique(colid[rows], return_inverse=True)
            per = max(200, a.coarsest_rows // nb)
            lv = mlr.build(Ab, ci[rows], cj[rows], ck[rows], cid, coarsest_rows=per)
            bands.append(dict(rows=rows, levels=lv) if own is None else dict(rows=rows, levels=lv, own=own))
        # global coarsest operator: P0^T T P0
        blocks, off = [], [0]
        for bd in bands:
            P0 = None
            for lv in bd["levels"][:-1]:
                P0 = lv.P if P0 is None else (P0 @ lv.P)
            if P0 is None:
                P0 = sp.identity(bd["rows"].size, format="csr")
            blocks.append(P0)
            off.append(off[-1] + P0.shape[1])
        inv = None
        if a.overlap == 0:
            P0 = sp.block_diag(blocks, format="csr")
            A0 = (P0.T @ T @ P0).toarray()
            inv = np.linalg.inv(A0)
        t_setup = time.perf_counter() - t0
        sizes = [[lv.n for lv in bd["levels"]] for bd in bands]
        print(f"bands {nb}: setup {t_setup:.1f} s, level rows per band {sizes}, global coarsest {off[-1]}", flush=True)
        for label, glob in (("local hierarchies only", None), ("global coarsest level", (inv, off))):
            if glob is not None and (nb == 1 or a.overlap > 0):
                continue
            t0 = time.perf_counter()
            x, its, rr = fgmres(A, b, lambda r: band_cycle(bands, r, glob, refresh=a.refresh if a.overlap > 0 else 0), maxit=a.maxit)
            print(f"  {label:28s}: {its:4d} iterations, relres {rr:.2e}, {time.perf_counter() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
