"""The N > 1 path on CPU: partition rule, halo plan (host part of nkp_create_dist) and the
exchange protocol, with world_size 2 and 3 over gloo; plus an in-process sweep over P = 1..8."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ora
from nk_ocn_tracer_jacobian_precond_amd import dist as nd
from nk_ocn_tracer_jacobian_precond_amd import solver, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, mode, out, extra=(), env_extra=None):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), "--mode", mode, "--out", out, *extra],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_halo_plan_and_exchange_over_gloo(tmp_path, world):
    res = launch(world, "cpu-plan", str(tmp_path / "plan"))
    assert all(r["spmv_bit_exact"] for r in res)
    assert sum(r["m_loc"] for r in res) > 0
    # latitude bands: every rank talks to at most its two neighbours
    assert all(r["neighbours"] <= 2 for r in res)


@pytest.mark.parametrize("P", [1, 2, 3, 8])
def test_partition_and_plan_in_process(P):
    """Pin p6 (host part): for every rank the remapped local SpMV on [own | halo] equals the global one."""
    p = synth.generate(imt=24, jmt=20, km=10, adv="upwind3", hmix="isop", seed=5)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    starts = nd.snap_partition(blk, P)
    assert starts[0] == 0 and starts[-1] == p.flat_len and np.all(np.diff(starts) >= 0)
    assert set(starts.tolist()) <= set(blk.tolist())                      # cuts sit on water-column boundaries
    x = np.random.default_rng(0).standard_normal(p.flat_len)
    y_ref = ora.spmv(p.rowptr, p.colind, p.nzval, x)
    total_need = 0
    for r in range(P):
        loc = nd.local_slice(p.rowptr, p.colind, p.nzval, blk, starts, r)
        f, m = loc["fst_row"], loc["m_loc"]
        ext, halo, need = nd.plan_host(loc["rowptr"], loc["colind"], starts, r)
        assert need[r] == 0 and need.sum() == halo.size
        assert np.all(np.diff(halo) > 0)
        owners = np.searchsorted(starts, halo, side="right") - 1
        assert np.array_equal(np.bincount(owners, minlength=P), need)
        y = ora.spmv(loc["rowptr"], ext, loc["val"], np.concatenate([x[f:f + m], x[halo]]))
        assert np.array_equal(y, y_ref[f:f + m])
        total_need += halo.size
    assert (total_need == 0) == (P == 1)


def test_plan_rejects_bad_input():
    with pytest.raises(solver.NkpError):
        nd.plan_host(np.array([0, 1], np.int32), np.array([7], np.int32), np.array([0, 1, 2], np.int64), 0)   # column 7 >= n_global 2


@pytest.mark.parametrize("cnt", [2, 3, 4])
def test_tracer_per_rank_partition_in_process(cnt):
    """Weak-scaling partition: rank t owns tracer t of a cnt-tracer coupled system, built from the single-tracer
    problem alone.  Its rows equal the corresponding rows of the full coupled matrix, the halo is every other
    tracer's copy of every cell, and the remapped local SpMV reproduces the global one bit for bit."""
    p1 = synth.generate(imt=16, jmt=14, km=8, adv="upwind3", hmix="isop", seed=4)
    pc = synth.generate(imt=16, jmt=14, km=8, adv="upwind3", hmix="isop", seed=4, coupled_tracer_cnt=cnt)
    tsl = p1.tracer_state_len
    x = np.random.default_rng(0).standard_normal(pc.flat_len)
    y_ref = ora.spmv(pc.rowptr, pc.colind, pc.nzval, x)
    for r in range(cnt):
        loc, starts, n_global = nd.tracer_slice(p1, r, cnt)
        assert n_global == pc.flat_len and starts[r] == r * tsl
        lo, hi = pc.rowptr[r * tsl], pc.rowptr[(r + 1) * tsl]
        assert np.array_equal(loc["rowptr"], pc.rowptr[r * tsl:(r + 1) * tsl + 1] - lo)
        assert np.array_equal(loc["colind"], pc.colind[lo:hi]) and loc["val"].tobytes() == pc.nzval[lo:hi].tobytes()
        ext, halo, need = nd.plan_host(loc["rowptr"], loc["colind"], starts, r)
        assert halo.size == (cnt - 1) * tsl and np.all(need[np.arange(cnt) != r] == tsl)
        y = ora.spmv(loc["rowptr"], ext, loc["val"], np.concatenate([x[r * tsl:(r + 1) * tsl], x[halo]]))
        assert np.array_equal(y, y_ref[r * tsl:(r + 1) * tsl])


def test_tracer_partition_exchange_over_gloo(tmp_path):
    res = launch(2, "cpu-plan", str(tmp_path / "plan_t"), extra=("--partition", "tracers"))
    assert all(r["spmv_bit_exact"] for r in res)
    assert all(r["neighbours"] == 1 and r["n_halo"] == r["m_loc"] for r in res)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config3_partition_keeps_tracers_and_columns_whole(world):
    """bench.py's default N > 1 layout (BASELINE configs[3]: 4 coupled tracers, tracer-major rows, the reference's
    contiguous row-block rule snapped to water columns): a rank holds whole tracers (world <= 4) or a band of ONE
    tracer (world = 8), never a split column; the halo plan of every rank addresses the other tracers' copies of its cells."""
    import numpy as np
    from nk_ocn_tracer_jacobian_precond_amd import dist as nd
    from nk_ocn_tracer_jacobian_precond_amd import solver, synth
    p = synth.generate(imt=24, jmt=20, km=8, adv="upwind3", hmix="isop", seed=4, coupled_tracer_cnt=4)
    tsl = p.tracer_state_len
    blk = solver.column_blocks(p.col_start(), tsl, 4)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 4)
    starts = nd.snap_partition(blk, world)
    assert starts[0] == 0 and starts[-1] == p.flat_len and np.all(np.diff(starts) > 0)
    for r in range(world):
        loc = nd.local_slice(p.rowptr, p.colind, p.nzval, blk, starts, r, ci, cj)       # raises if a column is cut
        f, m = loc["fst_row"], loc["m_loc"]
        tracers = {f // tsl, (f + m - 1) // tsl}
        if world <= 4:
            assert f % tsl == 0 and m == (4 // world) * tsl                                # whole tracers
        else:
            assert len(tracers) == 1                                                       # a band of one tracer
        ext, halo, need = nd.plan_host(loc["rowptr"], loc["colind"], starts, r)
        own_cells = np.unique(np.arange(f, f + m) % tsl)
        other = halo[(halo // tsl) != (f // tsl)] if world >= 4 else halo
        assert np.isin(other % tsl, own_cells).all() or world == 8                         # same-cell couplings only (bands add lateral halo)


def test_cell_major_order_and_slices():
    """SURVEY.md section 8e-2 (host part): the cell-major renumbering of a tracer-major coupled system is a permutation that
    keeps water columns whole and puts the tracers of one cell next to each other; the rank slices of the permuted matrix,
    stacked, are P A P^T; and a band of cells needs only the band edge from other ranks, not the other tracers' vectors."""
    import scipy.sparse as sp
    cnt = 3
    p = synth.generate(imt=24, jmt=20, km=10, adv="upwind3", hmix="isop", seed=5, coupled_tracer_cnt=cnt)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, cnt)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), cnt)
    perm, inv, blk_new, col_t, col_src = solver.cell_major_order(blk, cnt)
    n = p.flat_len
    assert np.array_equal(np.sort(perm), np.arange(n)) and np.array_equal(perm[inv], np.arange(n))
    assert np.array_equal(col_t, np.tile(np.arange(cnt), (blk.size - 1) // cnt))
    assert np.array_equal(np.diff(blk_new), np.diff(blk)[col_src])                 # whole columns, same lengths
    per = (blk.size - 1) // cnt
    assert np.array_equal(col_src.reshape(per, cnt), np.arange(per)[:, None] + per * np.arange(cnt)[None, :])
    A = p.scipy_csr()
    want = A[perm][:, perm].tocsr()
    want.sort_indices()
    P = 3
    parts, halo_total, m_total = [], 0, 0
    for r in range(P):
        loc, starts, perm_r = nd.cell_major_slice(p.rowptr, p.colind, p.nzval, blk, cnt, P, r, ci, cj)
        assert np.array_equal(perm_r, perm)
        assert set(starts.tolist()) <= set(blk_new[::cnt].tolist())              # cuts between cells
        assert np.array_equal(loc["col_t"], col_t[np.searchsorted(blk_new, starts[r]):np.searchsorted(blk_new, starts[r + 1])])
        parts.append(sp.csr_matrix((loc["val"], loc["colind"], loc["rowptr"]), shape=(loc["m_loc"], n)))
        ext, halo, need = nd.plan_host(loc["rowptr"], loc["colind"], starts, r)
        halo_total += halo.size
        m_total += loc["m_loc"]
        # grid positions travel with the columns
        assert np.array_equal(loc["col_i"], np.asarray(ci)[col_src][np.searchsorted(blk_new, starts[r]):np.searchsorted(blk_new, starts[r + 1])])
    got = sp.vstack(parts).tocsr()
    assert (got != want).nnz == 0
    for r in range(got.shape[0]):
        row = got.indices[got.indptr[r]:got.indptr[r + 1]]
        assert np.all(np.diff(row) > 0)
    assert m_total == n
    # tracer-major blocks of the same system: every rank would need (cnt - 1) / cnt of the whole vector
    tracer_major_halo = 0
    st = nd.snap_partition(blk, P)
    for r in range(P):
        loc = nd.local_slice(p.rowptr, p.colind, p.nzval, blk, st, r)
        tracer_major_halo += nd.plan_host(loc["rowptr"], loc["colind"], st, r)[1].size
    assert halo_total < 0.5 * tracer_major_halo, (halo_total, tracer_major_halo)


@pytest.mark.parametrize("world,partition", [(2, "bands"), (3, "bands"), (2, "cells"), (3, "cells"), (2, "tracers")])
def test_overlap_plan_over_gloo(tmp_path, world, partition):
    """Everything nkp_create_dist decides on the host, on CPU over gloo (nkp_dist_overlap_plan_host): the halo completed to
    whole water columns keeps the distributed SpMV bit-exact; the overlap picks exactly the lateral neighbours (every halo
    column of a band cut, none of the other tracers' columns of a tracer cut); and the matrix the rank's hierarchy is built
    from is the global matrix restricted to [own rows | overlap rows], entry for entry."""
    res = launch(world, "cpu-overlap-plan", str(tmp_path / "oplan"), extra=("--partition", partition))
    assert all(r["halo_complete"] and r["spmv_bit_exact"] for r in res), res
    assert all(r["ras"] == r["ras_expected"] for r in res), res
    if partition == "tracers":
        assert all(r["ras"] == 0 and r["n_sel"] == 0 for r in res), res
    else:
        assert all(r["ras"] == 1 for r in res) and sum(r["n_sel"] for r in res) > 0, res
        assert all(r["sel_rows_ok"] and r["ext_matrix_ok"] and r["blocks_ok"] and r["coords_ok"] and r["tracers_ok"] for r in res), res


@pytest.mark.parametrize("world", [2, 3])
def test_plan_failure_on_one_rank_ends_every_rank(tmp_path, world):
    """A rank-local failure inside nkp_create_dist's host plan (here: one rank holds a column index outside the matrix) must
    not strand the other ranks in the next exchange: the plan agrees on success after every rank-local check, the failing
    rank returns its own error (NKP_EINVAL) and every peer NKP_ECOMM naming it.  The launch itself has a deadline: a hang
    fails the test."""
    res = launch(world, "cpu-plan-fail", str(tmp_path / "pfail"))
    bad = res[0]["bad_rank"]
    for r in res:
        if r["rank"] == bad:
            assert r["code"] == -1 and "out of range" in r["message"], r
        else:
            assert r["code"] == -5 and f"rank {bad} failed" in r["message"], r


@pytest.mark.parametrize("world,partition", [(8, "bands"), (8, "cells")])
def test_overlap_plan_world_8(tmp_path, world, partition):
    """The driver's scaling run goes to 8 ranks: the same host-side plan checks as above with 8 processes -- configs[4]'s
    latitude bands and configs[3]'s cell-major bands of a coupled system (a 2-tracer stand-in on a grid with 8 bands of
    at least three latitude rows)."""
    res = launch(world, "cpu-overlap-plan", str(tmp_path / "oplan8"), extra=("--partition", partition, "--grid", "24x40x8"))
    assert all(r["halo_complete"] and r["spmv_bit_exact"] for r in res), res
    assert all(r["ras"] == 1 for r in res) and sum(r["n_sel"] for r in res) > 0, res
    assert all(r["sel_rows_ok"] and r["ext_matrix_ok"] and r["blocks_ok"] and r["coords_ok"] and r["tracers_ok"] for r in res), res
