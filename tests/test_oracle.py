"""Pin the CPU oracle against the committed SuperLU (SciPy) fixtures before anything trusts it."""
import numpy as np
import pytest

import oracle_binding as ora


def test_spmv_known_answer(golden):
    y = ora.spmv(golden.rowptr, golden.colind, golden.val, golden.gold["x_test"])
    ref = golden.gold["y_spmv"]
    assert np.allclose(y, ref, rtol=1e-13, atol=1e-13 * np.abs(ref).max())


def test_direct_solve_matches_superlu(golden):
    """Reference contract: B in, X out, berr ~ machine eps (pdgssvx_ABglobal with IterRefine)."""
    for g in golden.groups():
        b = golden.rhs(g)
        x, berr = ora.direct_solve(golden.rowptr, golden.colind, golden.val, b)
        xg = golden.gold["x_" + g]
        assert np.linalg.norm(x - xg) / np.linalg.norm(xg) < 1e-12
        assert berr < 1e-15
        assert abs(ora.berr(golden.rowptr, golden.colind, golden.val, xg, b) - golden.gold["berr_" + g]) < 1e-16


def test_column_blocks_exact(golden):
    """The banded LU of each water column block is exact: A_blk z = r block by block."""
    bw, nodiag, maxlen = ora.colblock_measure(golden.rowptr, golden.colind, golden.val, golden.blk_start)
    assert nodiag == 0 and maxlen <= golden.km
    assert bw == (2 if golden.name.startswith("penta") else 1)
    P = 1 if bw <= 1 else 2
    fac, dropped = ora.colblock_factor(golden.rowptr, golden.colind, golden.val, golden.blk_start, P)
    assert dropped == 0
    r = np.random.default_rng(5).standard_normal(golden.n)
    z = ora.colblock_apply(golden.n, golden.blk_start, P, fac, r)
    import scipy.sparse as sp
    A = sp.csr_matrix((golden.val, golden.colind, golden.rowptr), shape=(golden.n, golden.n)).tocoo()
    blk = np.searchsorted(golden.blk_start, np.arange(golden.n), side="right") - 1
    m = blk[A.row] == blk[A.col]
    Bd = sp.csr_matrix((A.data[m], (A.row[m], A.col[m])), shape=A.shape)
    assert np.linalg.norm(Bd @ z - r) / np.linalg.norm(r) < 1e-12


def test_fgmres_port_converges_to_superlu(golden):
    for g in golden.groups():
        b = golden.rhs(g)
        x, info = ora.fgmres(golden.rowptr, golden.colind, golden.val, golden.blk_start, b, restart=120, rtol=1e-12, max_iters=3000)
        assert info["status"] == 0, info
        xg = golden.gold["x_" + g]
        assert np.linalg.norm(x - xg) / np.linalg.norm(xg) < 1e-8


def test_multi_dot(golden):
    rng = np.random.default_rng(2)
    V = rng.standard_normal((5, golden.n))
    w = rng.standard_normal(golden.n)
    out = ora.multi_dot(V, w)
    assert np.allclose(out[:5], V @ w, rtol=1e-13) and np.isclose(out[5], w @ w, rtol=1e-13)
