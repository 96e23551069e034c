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
