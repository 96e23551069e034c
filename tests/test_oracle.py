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


# ---------------------------------------------------------------- the C / OpenMP port of the multilevel cycle (bench.py's cpu_baseline)
def _same_partition(a, b):
    a, b = np.asarray(a, np.int64), np.asarray(b, np.int64)
    pairs = np.unique(a * (int(b.max()) + 1) + b)
    return a.size == b.size and pairs.size == np.unique(a).size == np.unique(b).size


@pytest.mark.parametrize("grid,refine,k33", [((24, 20, 10), 1.0, True), ((40, 46, 20), 12.0, True), ((40, 46, 20), 1.0, False)])
def test_ml_oracle_matches_scipy_restatement(grid, refine, k33):
    """oracle/ml_oracle.c against tests/ml_reference.py: same levels, same coarse cells and columns, same V(3,3) cycle to
    rounding, and FGMRES around it reaches 1e-10 in the restatement's iteration count +- 2."""
    import ml_reference as mlr
    import scipy.sparse.linalg as spla
    from nk_ocn_tracer_jacobian_precond_amd import solver, synth
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=2, u_scale=3.0 * refine,
                       ah=4.0e6 * refine ** 2, isop_k33=k33)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    colid = np.cumsum(p.ind_k == 0) - 1
    A = p.scipy_csr()
    levels = mlr.build(A, p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64), colid, coarsest_rows=300)
    M = ora.MlOracle(p.rowptr, p.colind, p.nzval, blk, ci, cj, coarsest_rows=300)
    assert [r for r, _ in M.levels()] == [lv.n for lv in levels]
    assert [z for _, z in M.levels()] == [lv.A.nnz for lv in levels]
    to_ref = np.arange(p.flat_len)
    for l in range(len(levels) - 1):
        cmap, _ = M.maps(l)
        ref_of_fine = levels[l].cmap[to_ref]
        assert _same_partition(cmap, ref_of_fine), f"coarse cells differ on level {l}"
        nxt = np.empty(levels[l + 1].n, np.int64)
        nxt[cmap] = ref_of_fine
        to_ref = nxt
        assert _same_partition(M.maps(l + 1)[1], levels[l].coarse_colid[to_ref]), f"coarse columns differ on level {l + 1}"
    r = np.random.default_rng(3).standard_normal(p.flat_len)
    z, z_ref = M.apply(r), mlr.cycle(levels, 0, r)
    assert np.linalg.norm(z - z_ref) <= 1e-9 * np.linalg.norm(z_ref)
    b = np.random.default_rng(1).standard_normal(p.flat_len)
    x, info = M.fgmres(b, rtol=1e-10)
    assert info["status"] == 0 and np.linalg.norm(b - A @ x) <= 1.0001e-10 * np.linalg.norm(b)
    its = [0]
    Mop = spla.LinearOperator(A.shape, matvec=lambda v: mlr.cycle(levels, 0, np.asarray(v, np.float64)), dtype=np.float64)
    spla.gmres(A, b, M=Mop, rtol=1e-10, restart=200, maxiter=5, callback=lambda rr: its.__setitem__(0, its[0] + 1), callback_type="pr_norm")
    assert abs(info["iters"] - its[0]) <= 2, (info, its)
    M.close()
