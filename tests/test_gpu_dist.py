"""Distributed solve with two ranks sharing the one GPU of the test box (gloo + host staging of the
collectives; on a multi-GPU node the same code runs with backend nccl = RCCL, one GPU per rank)."""
import pytest

from test_dist_gloo import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_solve(tmp_path, world):
    res = launch(world, "gpu-solve", str(tmp_path / "solve"), extra=("--grid", "40x46x20"))
    assert all(r["spmv_bit_exact"] for r in res), res
    assert all(r["self_test"] for r in res), res                # the transport pre-flight bench.py runs (comm_self_test)
    assert all(not r["comm_errors"] for r in res), res
    assert all(r["status"] == 0 and r["relres"] <= 1e-10 for r in res), res
    assert res[0]["relres_checked"] <= 1.1e-10
    assert len({r["iters"] for r in res}) == 1                  # every rank took the same global decisions
    # latitude bands: most rows have no off-rank column, their SpMV runs while the halo travels on a second stream
    assert all(r["dist_overlap"] == 1 and r["interior_rowblocks"] > 0 for r in res), res
    # and every rank's hierarchy covers one ring of its neighbours' water columns (restricted additive Schwarz)
    assert all(r["ras"] == 1 and r["ras_rows"] > 0 for r in res), res
    plain = launch(world, "gpu-solve", str(tmp_path / "solve_plain"), extra=("--grid", "40x46x20"), env_extra={"NKP_DIST_RAS": "0"})
    assert all(r["ras"] == 0 and r["status"] == 0 for r in plain), plain
    assert res[0]["iters"] <= plain[0]["iters"], (res[0]["iters"], plain[0]["iters"])     # the overlap is there to save iterations


def test_one_allreduce_per_arnoldi_step(tmp_path):
    """nkp_tuning.dist_one_reduce (opt-in): the norm of the orthogonalised vector comes out of the reduced multi-dot message
    (w.w - sum h^2) instead of a second allreduce.  It converges to the same tolerance; the iterations it costs (the identity
    assumes an orthonormal basis, single-pass Gram-Schmidt keeps it only approximately) are why it is not the default."""
    two = launch(2, "gpu-solve", str(tmp_path / "two"), extra=("--grid", "40x46x20"))
    one = launch(2, "gpu-solve", str(tmp_path / "one"), extra=("--grid", "40x46x20"), env_extra={"NKP_DIST_ONE_REDUCE": "1"})
    for res in (one, two):
        assert all(r["status"] == 0 and r["relres"] <= 1e-10 and not r["comm_errors"] for r in res), res
        assert res[0]["relres_checked"] <= 1.1e-10
    assert two[0]["iters"] - 2 <= one[0]["iters"] <= two[0]["iters"] * 1.25 + 2, (one[0]["iters"], two[0]["iters"])


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
    # every row couples to the other tracers' copy of its cell: no interior rows, nothing to overlap
    assert all(r["dist_overlap"] == 0 for r in res), res
    # the other tracers' columns sit at positions this rank owns: they are not lateral neighbours, the hierarchy stays rank-local
    assert all(r["ras"] == 0 and r["ras_rows"] == 0 for r in res), res


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


@pytest.mark.parametrize("world,case", [(2, "penta_12x10x6"), (3, "pair_8x8x5")])
def test_solve_ABdist_cli_multi_process(tmp_path, golden_by_name, world, case):
    """bin/solve_ABdist as a real multi-process program (reference src/solve_ABdist.c:115-244, 334-418): `world`
    processes, each with its own row block (the reference's n/P rule snapped to water columns), halo exchange and
    allreduce through the host-staged file transport (the box has one GPU; RCCL needs one per rank), slices gathered
    on rank 0, which writes the tracer file.  Result against the SuperLU fixture; the id file of an EARLIER job with
    another tag lying on the path must not be picked up (covered by the single-rank RCCL test's tag)."""
    import os
    import shutil
    import subprocess
    import numpy as np
    from nk_ocn_tracer_jacobian_precond_amd import nc3
    g = golden_by_name(case)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "nk_ocn_tracer_jacobian_precond_amd", "bin", "solve_ABdist")
    dst = str(tmp_path / "B_dist.nc")
    shutil.copy(g.tracer_path, dst)
    comm_dir = tmp_path / "comm"
    comm_dir.mkdir()
    names = ",".join(g.varnames)
    procs = []
    for r in range(world):
        env = dict(os.environ, NKP_COMM="file", NKP_COMM_DIR=str(comm_dir), NKP_COMM_TIMEOUT="60", NKP_RTOL="1e-12", RANK=str(r),
                   WORLD_SIZE=str(world), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([exe, "-D1", "-n", "1", "-v", names, g.matrix_path, dst], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se + so
    assert sum("nkp_create_dist: rows [" in so for so, _ in outs) == world
    rows = sorted(int(so.split("nkp_create_dist: rows [")[1].split(",")[0]) for so, _ in outs)
    assert rows[0] == 0 and len(set(rows)) == world            # every rank owned a different row block
    # a coupled pair over several ranks is solved in cell-major order (bands of whole cells) and written back tracer-major
    assert all(("cell-major order" in so) == (g.cnt > 1) for so, _ in outs)
    out = nc3.NcFile(dst)
    for grp in g.groups():
        k = g.varnames.index(grp)
        x = np.concatenate([out.get(v)[g.ind_k, g.ind_j, g.ind_i] for v in g.varnames[k:k + g.cnt]])
        ref = g.gold["x_" + grp]
        assert np.linalg.norm(x - ref) / np.linalg.norm(ref) <= 1e-7
    assert not list(comm_dir.iterdir())                         # the transport cleaned up after itself


def test_stale_rccl_id_file_is_ignored(tmp_path, golden_by_name):
    """An id file left on the path by another job (different tag) must not be taken: the single-rank RCCL run publishes
    and removes its own."""
    import os
    import shutil
    import subprocess
    g = golden_by_name("tri_12x10x6")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "nk_ocn_tracer_jacobian_precond_amd", "bin", "solve_ABdist")
    dst = str(tmp_path / "B.nc")
    shutil.copy(g.tracer_path, dst)
    idfile = tmp_path / "rccl.id"
    idfile.write_bytes(b"other-job".ljust(64, b"\0") + bytes(128))
    env = dict(os.environ, NKP_FORCE_DIST="1", NKP_RCCL_ID_FILE=str(idfile), NKP_JOB_ID="job-42", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([exe, "-v", "IAGE", g.matrix_path, dst], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr + r.stdout
    assert not idfile.exists()                                  # rank 0 removed its id once the communicator existed


@pytest.mark.parametrize("world", [2, 3])
def test_distributed_solve_cell_major_bands(tmp_path, world):
    """a 2-tracer coupled system renumbered cell-major and cut into bands of whole cells -- the same-cell
    couplings are rank-local (so is their share of the hierarchy: col_t keeps the tracers apart), the band edge is overlap."""
    res = launch(world, "gpu-solve", str(tmp_path / "solve_c"), extra=("--grid", "40x46x20", "--partition", "cells"))
    assert all(r["spmv_bit_exact"] for r in res), res
    assert all(not r["comm_errors"] for r in res), res
    assert all(r["status"] == 0 and r["relres"] <= 1e-10 for r in res), res
    assert res[0]["relres_checked"] <= 1.1e-10
    assert len({r["iters"] for r in res}) == 1
    assert all(r["dist_overlap"] == 1 and r["ras"] == 1 and r["ras_rows"] > 0 for r in res), res
    # the tracer-major cut of a coupled system (one tracer per rank) pays for ignoring the coupling in its preconditioner
    if world == 2:
        tm = launch(world, "gpu-solve", str(tmp_path / "solve_tm"), extra=("--grid", "40x46x20", "--partition", "tracers"))
        assert res[0]["iters"] <= tm[0]["iters"], (res[0]["iters"], tm[0]["iters"])


def test_bench_distributed_path_on_one_rank(tmp_path):
    """bench.py's N > 1 code path (configs[3] layout, library RCCL communicator after its pre-flight self-test, barriers,
    max-over-ranks timing) driven with a single rank: RCCL wants one GPU per rank and the test box has one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--grid", "40x46x20", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--rhs-batch", "0", "--round1-steps", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert "library RCCL communicator" in line["config"]["multi_gpu"], line["config"]
    assert "FALLBACK" not in line["config"]["multi_gpu"]
    assert "self-test" not in out.stderr, out.stderr[-2000:]
    assert line["value"] > 0 and max(line["solve"]["relres"]) <= 1e-10
    assert line["solve"]["relres_checked_with_torch"] <= 1.1e-10


def test_cell_major_order_on_one_gpu():
    """The hierarchy does not depend on tracer-major rows: the same coupled system renumbered cell-major (col_t names the
    tracer of every column) is solved to the same x in about the same number of iterations."""
    import numpy as np
    from nk_ocn_tracer_jacobian_precond_amd import solver, synth
    cnt = 2
    p = synth.generate(imt=40, jmt=46, km=20, adv="upwind3", hmix="isop", seed=2, coupled_tracer_cnt=cnt)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, cnt)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), cnt)
    b = np.random.default_rng(0).standard_normal(p.flat_len)
    s0 = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, coupled_tracer_cnt=cnt, col_i=ci, col_j=cj, rtol=1e-11)
    x0, i0 = s0.solve(b)
    s0.close()
    perm, inv, blk_new, col_t, col_src = solver.cell_major_order(blk, cnt)
    rp, cc, vv = solver.permuted_rows(p.rowptr, p.colind, p.nzval, perm, inv, 0, p.flat_len)
    s1 = solver.NkpSolver(rp, cc, vv, blk_new, coupled_tracer_cnt=cnt, col_i=np.asarray(ci)[col_src], col_j=np.asarray(cj)[col_src], col_t=col_t, rtol=1e-11)
    x1, i1 = s1.solve(b[perm])
    s1.close()
    assert i0["status"] == 0 and i1["status"] == 0
    assert np.linalg.norm(x1[inv] - x0) <= 1e-8 * np.linalg.norm(x0)
    assert abs(i1["iters"] - i0["iters"]) <= max(3, 0.2 * i0["iters"]), (i0["iters"], i1["iters"])


def test_transport_self_test_catches_a_bad_transport(tmp_path):
    """dist.comm_self_test, the pre-flight bench.py runs before it builds on the library's RCCL communicator: passes on a
    sound transport, raises on one whose allreduce returns without reducing, and reports (instead of hanging with it) one
    whose call does not come back within the deadline."""
    res = launch(2, "gpu-selftest", str(tmp_path / "selftest"))
    assert all(r["sound"] and r["sound_again"] for r in res), res
    assert all(r["lying"].startswith("caught") and r["hanging"].startswith("caught") for r in res), res
