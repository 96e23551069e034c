"""Host-side aggregation of the multilevel preconditioner (nkp_ml_plan_host, csrc/multilevel.hip) against the
independent scipy restatement tests/ml_reference.py: identical coarse cells and coarse columns on every level.
No GPU needed: the plan is host code."""
import numpy as np
import pytest

import ml_reference as mlr
from nk_ocn_tracer_jacobian_precond_amd import solver, synth


def _same_partition(a, b):
    """two labelings of the same items describe the same partition"""
    a, b = np.asarray(a, np.int64), np.asarray(b, np.int64)
    if a.size != b.size:
        return False
    pairs = np.unique(a * (int(b.max()) + 1) + b)
    return pairs.size == np.unique(a).size == np.unique(b).size


@pytest.mark.parametrize("grid,refine,k33", [((24, 20, 10), 1.0, False), ((40, 46, 20), 1.0, False), ((40, 46, 20), 12.0, True),
                                            ((64, 60, 30), 12.0, False)])
def test_split_aggregation_matches_restatement(grid, refine, k33):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=2, u_scale=3.0 * refine,
                       ah=4.0e6 * refine ** 2, isop_k33=k33)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    colid = np.cumsum(p.ind_k == 0) - 1
    levels = mlr.build(p.scipy_csr(), p.ind_i.astype(np.int64), p.ind_j.astype(np.int64), p.ind_k.astype(np.int64), colid)
    rows, cmaps, colofs = solver.ml_plan_host(p.rowptr, p.colind, p.nzval, blk, ci, cj)
    assert list(rows) == [lv.n for lv in levels]
    # the product numbers coarse rows differently: compare through the fine level
    to_ref = np.arange(p.flat_len)          # product row of level l -> restatement row of level l
    for l in range(len(levels) - 1):
        ref_of_fine = levels[l].cmap[to_ref]                      # restatement's coarse row of every product row
        assert _same_partition(cmaps[l], ref_of_fine), f"coarse cells differ on level {l}"
        nxt = np.empty(rows[l + 1], np.int64)
        nxt[cmaps[l]] = ref_of_fine
        to_ref = nxt
        assert _same_partition(colofs[l], levels[l].coarse_colid[to_ref]), f"coarse columns differ on level {l + 1}"
