"""Distributed solve with two ranks sharing the one GPU of the test box (gloo + host staging of the
collectives; on a multi-GPU node the same code runs with backend nccl = RCCL, one GPU per rank)."""
import pytest

from test_dist_gloo import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_solve(tmp_path, world):
    res = launch(world, "gpu-solve", str(tmp_path / "solve"), extra=("--grid", "40x46x20"))
    assert all(r["spmv_bit_exact"] for r in res), res
    assert all(not r["comm_errors"] for r in res), res
    assert all(r["status"] == 0 and r["relres"] <= 1e-10 for r in res), res
    assert res[0]["relres_checked"] <= 1.1e-10
    assert len({r["iters"] for r in res}) == 1                  # every rank took the same global decisions


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_solve_one_tracer_per_rank(tmp_path, world):
    """Weak-scaling layout: rank t holds tracer t of a `world`-tracer coupled system (its preconditioner ignores the
    coupling to the other tracers, the Krylov operator does not)."""
    res = launch(world, "gpu-solve", str(tmp_path / "solve_t"), extra=("--grid", "40x46x20", "--partition", "tracers"))
    assert all(r["spmv_bit_exact"] for r in res), res
    assert all(not r["comm_errors"] for r in res), res
    assert all(r["status"] == 0 and r["relres"] <= 1e-10 for r in res), res
    assert res[0]["relres_checked"] <= 1.1e-10
    assert len({r["iters"] for r in res}) == 1


def test_solve_ABdist_cli_with_builtin_rccl(tmp_path, golden_by_name):
    """The executable's distributed entry point with the library's own RCCL communicator (one rank: the
    multi-rank launch needs one GPU per rank, which the test box does not have)."""
    import os
    import shutil
    import subprocess
    import numpy as np
    from nk_ocn_tracer_jacobian_precond_amd import nc3
    g = golden_by_name("penta_12x10x6")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "nk_ocn_tracer_jacobian_precond_amd", "bin", "solve_ABdist")
    dst = str(tmp_path / "B_dist.nc")
    shutil.copy(g.tracer_path, dst)
    env = dict(os.environ, NKP_FORCE_DIST="1", NKP_RCCL_ID_FILE=str(tmp_path / "rccl.id"), NKP_RTOL="1e-12", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([exe, "-D1", "-n", "1", "-v", "IAGE", g.matrix_path, dst], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "nkp_create_dist: rows [0," in r.stdout
    x = nc3.NcFile(dst).get("IAGE")[g.ind_k, g.ind_j, g.ind_i]
    ref = g.gold["x_IAGE"]
    assert np.linalg.norm(x - ref) / np.linalg.norm(ref) <= 1e-7
