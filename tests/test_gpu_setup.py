"""The hierarchy the setup KERNELS build (csrc/mlsetup.hip: low-order twin, connectivity-aware coarse cells, Galerkin
products, colour-major operators, transfer maps) against the one the host routines of csrc/multilevel.hip build -- the
routines tests/test_ml_plan.py pins to the scipy restatement without a GPU.  Entry for entry, bit for bit; then the same
right-hand side through both preconditioners.  NKP_ML_DEVICE_MIN is the smallest level (rows) the kernels take."""
import numpy as np
import pytest

from nk_ocn_tracer_jacobian_precond_amd import solver, synth

pytestmark = pytest.mark.gpu

ARRAYS = ("rowptr", "colind", "valf", "val", "cmap", "rptr", "ridx", "blk_start", "fac", "perm0", "coarse_inv")


def _problem(grid, refine, k33, cnt=1, seed=2):
    p = synth.generate(imt=grid[0], jmt=grid[1], km=grid[2], adv="upwind3", hmix="isop", seed=seed, u_scale=3.0 * refine,
                       ah=4.0e6 * refine ** 2, isop_k33=k33, coupled_tracer_cnt=cnt)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, cnt)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), cnt)
    return p, blk, ci, cj


def _hierarchy(p, blk, ci, cj, cnt, dev_min, monkeypatch, **kw):
    monkeypatch.setenv("NKP_ML_DEVICE_MIN", str(dev_min))
    s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, coupled_tracer_cnt=cnt, col_i=ci, col_j=cj, precond=solver.PRECOND_MULTILEVEL, **kw)
    levels = s.get_int("levels")
    arrays = [{a: s.ml_level_array(l, a) for a in ARRAYS} for l in range(levels)]
    return s, arrays


@pytest.mark.parametrize("grid,refine,k33,cnt", [((24, 20, 10), 1.0, False, 1), ((40, 46, 20), 1.0, True, 1), ((40, 46, 20), 12.0, True, 1),
                                                ((64, 60, 30), 12.0, False, 1), ((36, 30, 12), 1.0, True, 2), ((100, 116, 60), 1.0, True, 1)])
def test_device_built_hierarchy_equals_host_built(grid, refine, k33, cnt, monkeypatch):
    p, blk, ci, cj = _problem(grid, refine, k33, cnt)
    # small coarsest level so that even the small grids get several levels
    monkeypatch.setenv("NKP_ML_COARSEST_ROWS", "300")
    s_host, h_host = _hierarchy(p, blk, ci, cj, cnt, 1 << 40, monkeypatch)
    s_dev, h_dev = _hierarchy(p, blk, ci, cj, cnt, 0, monkeypatch)
    assert s_host.get_int("ml_levels_on_device") == 0
    assert s_dev.get_int("ml_levels_on_device") == len(h_dev) - 1 >= 2          # every level but the last one
    assert len(h_host) == len(h_dev)
    for l, (a, b) in enumerate(zip(h_host, h_dev)):
        for name in ARRAYS:
            assert a[name].shape == b[name].shape, (l, name, a[name].shape, b[name].shape)
            assert np.array_equal(a[name].view(np.uint8), b[name].view(np.uint8)), f"level {l}: {name} differs"
    r = np.random.default_rng(5).standard_normal(p.flat_len)
    assert np.array_equal(s_host.precond_apply(r), s_dev.precond_apply(r))
    # a mixed hierarchy (kernels above 2000 rows, host below) is the same hierarchy again
    s_mix, h_mix = _hierarchy(p, blk, ci, cj, cnt, 2000, monkeypatch)
    assert 0 < s_mix.get_int("ml_levels_on_device") or p.flat_len < 2000
    for l, (a, b) in enumerate(zip(h_host, h_mix)):
        for name in ARRAYS:
            assert np.array_equal(a[name].view(np.uint8), b[name].view(np.uint8)), f"mixed, level {l}: {name} differs"
    for s in (s_host, s_dev, s_mix):
        s.close()


def test_device_built_hierarchy_solves(monkeypatch):
    """end to end on the kernels' hierarchy: same iteration count and same bits as on the host-built one"""
    p, blk, ci, cj = _problem((100, 116, 60), 1.0, True)
    b = np.random.default_rng(1).standard_normal(p.flat_len)
    out = []
    for dev_min in (1 << 40, 0):
        monkeypatch.setenv("NKP_ML_DEVICE_MIN", str(dev_min))
        with solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, precond=solver.PRECOND_MULTILEVEL, rtol=1e-10) as s:
            x, info = s.solve(b)
            out.append((x, info["iters"], info["relres"], s.get_int("ml_levels_on_device")))
    assert out[0][3] == 0 and out[1][3] >= 2
    assert out[0][1] == out[1][1] and out[1][2] <= 1e-10
    assert np.array_equal(out[0][0], out[1][0])
