"""BASELINE.json configs[3] and configs[4] and the resolution regime behind them, through the C ABI on one MI355X.

The residual every assertion uses is recomputed by the CPU oracle's SpMV, not taken from the solver's own report."""
import numpy as np
import pytest

import oracle_binding as ora
from nk_ocn_tracer_jacobian_precond_amd import solver, synth

pytestmark = pytest.mark.gpu

# iteration bounds of the gen_A pipeline test: 1.3 x measured
GEN_A_BOUNDS = {"upwind3": 96, "cent": 260}          # measured 74 / 199 (gpurun_out/r3m/genA2.log)


def _solve(p, cnt=1, **kw):
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, cnt)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), cnt)
    b = np.random.default_rng(5).standard_normal(p.flat_len)
    with solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, coupled_tracer_cnt=cnt, **kw) as s:
        x, info = s.solve(b, raise_on_fail=False)
        info["levels"] = s.get_int("levels")
    res = b - ora.spmv(p.rowptr, p.colind, p.nzval, x)
    info["true_relres"] = float(np.linalg.norm(res) / np.linalg.norm(b))
    print(f"MEASURED n={p.flat_len} cnt={cnt} kw={kw} iters={info['iters']} status={info['status']} relres={info['true_relres']:.2e}")
    return info


# iteration bounds of this file: 1.3 x what round 3 measured (gpurun_out/r3i/configs.log), so that a regression shows
@pytest.mark.parametrize("refine,bound", [(1.0, 72), (4.0, 52), (12.0, 54)])          # measured 55 / 40 / 41
def test_cell_courant_number_of_finer_grids(refine, bound):
    """The 3 degree x 60 grid with the cell-level coefficients of a grid `refine` times finer (velocities x refine,
    lateral diffusivity x refine^2: cell Courant and diffusion numbers of 3, 0.75 and 0.25 degree).  Round 1 stalled
    here (the aggregates mixed unconnected water); the iteration count must now stay flat."""
    p = synth.generate(imt=100, jmt=116, km=60, adv="upwind3", hmix="isop", seed=0, u_scale=3.0 * refine, ah=4.0e6 * refine ** 2)
    info = _solve(p)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= bound, info


def test_cell_courant_number_round1_recipe():
    """Same regime with the round-1 synthetic recipe (isopycnal cross terms without the K33 term: an indefinite mixing
    tensor).  It did not converge at all in round 1 (stall at 7e-3 after 20 000 iterations with block-Jacobi, 0.2-0.3
    with the multilevel cycle at 0.25 degree); with connectivity-aware aggregates it does, slowly."""
    p = synth.generate(imt=100, jmt=116, km=60, adv="upwind3", hmix="isop", seed=0, u_scale=36.0, ah=4.0e6 * 144, isop_k33=False)
    info = _solve(p, max_iters=6000)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= 1800, info                          # measured 1392


def test_config_4tracer_1deg():
    """configs[3]: 1 degree x 60 levels x 4 coupled tracers (n = 16.9 M, nnz = 344 M) on one GPU."""
    p = synth.generate(imt=320, jmt=384, km=60, adv="upwind3", hmix="isop", seed=0, coupled_tracer_cnt=4)
    assert p.flat_len == 4 * p.tracer_state_len and p.flat_len > 16_000_000
    info = _solve(p, cnt=4, restart=100)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= 78, info                            # measured 60


def test_config_quarter_degree():
    """configs[4]: 0.25 degree x 80 levels (1440 x 720 x 80, n = 50.7 M, nnz = 893 M) on one GPU -- the system that did
    not converge in round 1."""
    p = synth.generate(imt=1440, jmt=720, km=80, adv="upwind3", hmix="isop", seed=0)
    assert p.flat_len > 50_000_000
    info = _solve(p, restart=60)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= 136, info                           # measured 105


def test_coarsest_level_without_dense_inverse(monkeypatch):
    """Rough bathymetry can leave a coarsest level too large for the dense inverse (thousands of pocket stubs): it is then
    relaxed with many sweeps of the column smoother instead of failing the setup.  Forced here by lowering the limit."""
    p = synth.generate(imt=100, jmt=116, km=60, adv="upwind3", hmix="isop", seed=0)
    ref = _solve(p)
    monkeypatch.setenv("NKP_ML_DENSE_MAX", "50")
    info = _solve(p)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= 2 * ref["iters"] + 10, (info, ref)


@pytest.mark.parametrize("adv,hmix,max_iters", [("donor", "isop", 54), ("centred", "isop", 228), ("upwind3", "const", 140), ("centred", "const", 307),
                                                ("none", "const", 60)])       # measured 41 / 175 / 107 / 236 / 46
def test_operator_families_at_3_degrees(adv, hmix, max_iters):
    """Every advection / lateral-mixing family the reference's gen_A offers (src/gen_A.c:170-216), on the 3 degree x 60 grid
    of BASELINE configs[1]: the solve must meet 1e-10 on the oracle-recomputed residual within a bound that documents how
    hard the family is for the low-order twin (centred advection, the reference's default, is the hardest)."""
    p = synth.generate(imt=100, jmt=116, km=60, adv=adv, hmix=hmix, seed=0)
    info = _solve(p, max_iters=3000)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= max_iters, info


def test_centred_advection_at_1_degree():
    """The reference's DEFAULT advection scheme (centred, src/gen_A.c:99) at the headline size, 1 degree x 60: the hardest family
    for the low-order twin (cell Peclet numbers of 30 fully upwinded)."""
    p = synth.generate(imt=320, jmt=384, km=60, adv="centred", hmix="isop", seed=0)
    info = _solve(p, max_iters=2000)
    assert info["status"] == 0 and info["true_relres"] <= 1e-10, info
    assert info["iters"] <= 410, info                           # round 2 measured 315


@pytest.mark.parametrize("adv,bound", [("upwind3", GEN_A_BOUNDS["upwind3"]), ("cent", GEN_A_BOUNDS["cent"])])
def test_gen_A_pipeline_at_3_degrees(tmp_path, adv, bound):
    """The reference's whole chain at the size of BASELINE configs[1] (3 degree x 60 levels): synthetic circulation file ->
    bin/gen_A (src/gen_A.c, both the scheme of test/test_gen_A.csh and its default, centred advection) -> matrix file -> the
    library on the GPU; the residual is recomputed by the CPU oracle on the CSR read back from the generated file."""
    import os
    import subprocess
    from nk_ocn_tracer_jacobian_precond_amd import circ, nc3
    BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nk_ocn_tracer_jacobian_precond_amd", "bin")
    F, fills = circ.make_circulation(100, 116, 60, seed=0)
    cpath, mpath = str(tmp_path / "circ.nc"), str(tmp_path / "matrix.nc")
    circ.write_circ_file(cpath, F, fills, nc_type="float32")
    (tmp_path / "gen_A.opt").write_text(f"circ_fname {cpath}\nadv_type {adv}\nl_adv_enforce_divfree 1\nhmix_type isop_file\nvmix_type file\n"
                                        "sink_type const_shallow 365.0 10.0e2\n")
    r = subprocess.run([os.path.join(BIN, "gen_A"), "-o", str(tmp_path / "gen_A.opt"), mpath], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = nc3.NcFile(mpath)
    rp, ci, val = m.get("rowptr"), m.get("colind"), m.get("nzval_row_wise")
    ii, jj, kk = (m.get(f"tracer_state_ind_to_{c}") for c in "ijk")
    n = len(rp) - 1
    assert n > 300_000
    col_start = np.concatenate([np.flatnonzero(kk == 0), [len(kk)]]).astype(np.int32)
    blk = solver.column_blocks(col_start, len(kk), 1)
    cci, ccj = solver.column_coords(ii, jj, col_start, 1)
    b = np.random.default_rng(1).standard_normal(n)
    with solver.NkpSolver(rp, ci, val, blk, col_i=cci, col_j=ccj, max_iters=3000) as s:
        x, info = s.solve(b, raise_on_fail=False)
    relres = float(np.linalg.norm(b - ora.spmv(rp, ci, val, x)) / np.linalg.norm(b))
    print(f"MEASURED gen_A 3 degree adv={adv} n={n} nnz={len(val)} iters={info['iters']} status={info['status']} relres={relres:.2e} berr={info['berr']:.2e}")
    if adv == "cent":
        # the centred matrix of this circulation file stops at 3e-10 in the 2-norm at a componentwise backward error of 1e-15
        # (rows of very different size: the same status the coupled pair of test_gen_A_to_solve_pipeline gets); the library says
        # so with NKP_OK_BERR instead of claiming the tolerance
        assert info["status"] in (0, 3) and relres <= 1e-9 and info["berr"] <= 1e-14, (info, relres)
    else:
        assert info["status"] == 0 and relres <= 1e-10, (info, relres)
    assert info["iters"] <= bound, info
