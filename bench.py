#!/usr/bin/env python3
"""bench.py -- the driver's benchmark contract for the Jacobian-preconditioner solve path.

A "step" is ONE preconditioned solve A x = b to ||b-Ax||/||b|| <= 1e-10 for a fresh right-hand
side that is already resident in HBM (the matrix and the preconditioner hierarchy were set up
once, like the reference's factor-once / solve-per-tracer loop, src/solve_ABglobal.c:349-409).
Workload at N=1: the 1 degree x 60 level single-tracer Jacobian BASELINE.json's metric is quoted
on (configs[2]); synthetic, built by nk_ocn_tracer_jacobian_precond_amd.synth with the stencil
of the reference's shipped job (upwind3 + isop + vmix + shallow sink, test/test_gen_A.csh:22-23).
Since round 2 the synthetic isopycnal mixing carries the K33 term of the Redi tensor (see synth.py; without
it the tensor is indefinite, which no ocean model produces); the round-1 recipe is still timed beside it
(`round1_recipe` in the JSON line, never as `value`).

N > 1 (one process per GPU, launched by torch.distributed.run, RCCL through torch.distributed), `--multi-gpu`:
  c4 (default)  BASELINE configs[3]: the 1 degree x 60 x 4-tracer coupled Jacobian (n = 16.9 M), renumbered cell-major
                 and split over the N ranks by the reference's contiguous row-block rule
                (src/solve_ABdist.c:141-144) snapped to whole cells: every rank holds a latitude band of all four tracers
                (--c4-order tracer: the reference's tracer-major rows, N = 2 -> two tracers per rank, 4 -> one, 8 -> half)
  weak          the N-tracer coupled system, one tracer per rank (per-GPU work = the N = 1 workload)
  strong        the single-tracer matrix (or --grid 1440x720x80 = configs[4]) cut into N latitude bands

  python bench.py [--gpus N] [--steps K] [--warmup W] [--grid IxJxK] [--adv ..] [--hmix ..] [--multi-gpu c4|weak|strong]

Prints ONE JSON line on rank 0.  value = unknowns solved per second over the whole job.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MEASURED_COPY_GBS = 6290.0          # measured float4 copy on the same part (same guide, table row "HBM3E peak BW")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", default="320x384x60")
    ap.add_argument("--adv", default="upwind3")
    ap.add_argument("--hmix", default="isop")
    ap.add_argument("--restart", type=int, default=200)
    ap.add_argument("--ml-smooth", type=int, default=3)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--max-iters", type=int, default=20000)
    ap.add_argument("--recipe", choices=["k33", "round1"], default="k33",
                    help="synthetic isopycnal mixing: with the K33 term of the Redi tensor (default) or the round-1 recipe without it")
    ap.add_argument("--round1-steps", type=int, default=2, help="N = 1 extra: solves timed on the round-1 recipe (0 disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--multi-gpu", choices=["c4", "weak", "strong"], default="c4",
                    help="N > 1: c4 = BASELINE configs[3], the 4-tracer 1-degree system row-partitioned over the N ranks; "
                         "weak = one coupled 1-degree tracer per GPU (N-tracer system); strong = the --grid matrix split into N latitude bands")
    ap.add_argument("--rhs-batch", type=int, default=4,
                    help="N = 1 extra (reported beside `value`, never as `value`): that many right-hand sides in flight at once "
                         "on clones of the solver (nkp_clone), the reference's RHS loop run concurrently; 0 disables")
    ap.add_argument("--force-dist", action="store_true", help="developer switch: run the distributed code path even with one rank")
    ap.add_argument("--c4-order", choices=["cell", "tracer"], default="cell",
                    help="--multi-gpu c4: renumber the coupled system cell-major before cutting it (default), or cut the reference's tracer-major rows")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl",
                    help="N > 1 collectives: rccl = the library's own RCCL communicator (C only, no Python per iteration; "
                         "torch.distributed just carries the unique id), torch = torch.distributed callbacks")
    return ap.parse_args()


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (the PMC passes are separate rocprofv3 runs, committed per round)"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_" + suffix)))
    if not found:
        raise FileNotFoundError(suffix)
    return found[-1]


def host_cpu_share():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands each job a
    share of its cores; OpenMP would otherwise start one spinning thread per core of the machine)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()])):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get("NKP_BENCH_CPU_THREADS")
    return int(env) if env else n


def cpu_complete_solve(p, blk, ci, cj, rtol, restart):
    """ONE complete solve of the bench workload itself on the host, with the SAME algorithm the GPU runs -- low-order twin,
    connectivity-aware 2 x 2 aggregation, Galerkin operators, 2-colour water-column Gauss-Seidel V(3,3) cycle, FGMRES --
    in its C / OpenMP restatement oracle/ml_oracle.c (TEST INFRASTRUCTURE used only as this timed CPU leg) on all host
    cores.  Measured, not extrapolated; the right-hand side is drawn like the GPU's (standard normal)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ora
    cores = host_cpu_share()
    ora.set_num_threads(cores)
    t0 = time.perf_counter()
    # coarsest_rows = 3000 (the library's value until round 3): the GPU now ends its hierarchy one level earlier with a dense
    # inverse of 7177 rows, a matrix-core job; on the host that level is cheaper iterated, so the CPU keeps the deeper hierarchy
    M = ora.MlOracle(p.rowptr, p.colind, p.nzval, blk, ci, cj, coarsest_rows=3000)
    t_setup = time.perf_counter() - t0
    b = np.random.default_rng(1).standard_normal(p.flat_len)
    t0 = time.perf_counter()
    x, info = M.fgmres(b, restart=restart, rtol=rtol)
    t_solve = time.perf_counter() - t0
    relres = float(np.linalg.norm(b - ora.spmv(p.rowptr, p.colind, p.nzval, x)) / np.linalg.norm(b))
    levels = M.levels()
    M.close()
    return dict(n=p.flat_len, nnz=p.nnz, setup_s=t_setup, solve_s=t_solve, iterations=info["iters"], status=info["status"], relres=relres,
                cores=cores, levels=len(levels), unknowns_per_s=p.flat_len / t_solve)


def rhs_batch_throughput(torch, s, R, steps, n):
    """The reference loops over its right-hand sides one at a time against one factorisation
    (src/solve_ABglobal.c:370-409).  Here R of them are in flight at once: R host threads, each driving its own
    clone (own work vectors + stream, shared matrix and hierarchy).  Same solves, same bits, better use of the GPU
    during the latency-bound parts of the cycle."""
    import threading
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4321)
    B = torch.randn((R, steps, n), dtype=torch.float64, device="cuda", generator=gen)
    X = torch.zeros((R, steps, n), dtype=torch.float64, device="cuda")
    handles = [s] + [s.clone() for _ in range(R - 1)]
    infos = [[] for _ in range(R)]
    errors = []

    def work(k):
        try:
            for j in range(steps):
                infos[k].append(handles[k].solve_device(B[k, j].data_ptr(), X[k, j].data_ptr()))
        except Exception as exc:                             # surfaced below; never lose a failure in a thread
            errors.append(repr(exc))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(k,)) for k in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for h in handles[1:]:
        h.close()
    if errors:
        return {"error": errors}
    return {"concurrent_rhs": R, "solves": R * steps, "value": R * steps * n / dt, "unit": "unknowns/s",
            "ms_per_solve_effective": dt / (R * steps) * 1e3,
            "iterations": [i["iters"] for row in infos for i in row],
            "max_relres": max(i["relres"] for row in infos for i in row)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world != 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the solve path has no CPU fallback)")
    backend = os.environ.get("NKP_BENCH_BACKEND", "nccl")      # "gloo": developer rehearsal of N ranks on fewer GPUs (host-staged collectives)
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if a.force_dist:
        os.environ["NKP_FORCE_DIST"] = "1"
        os.environ.setdefault("MASTER_PORT", "29511")
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from nk_ocn_tracer_jacobian_precond_amd import dist as nd
    from nk_ocn_tracer_jacobian_precond_amd import solver, synth

    def all_reduce(t, op=dist.ReduceOp.SUM):
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:                                            # gloo rehearsal: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    def all_gather(parts, mine):
        if backend == "nccl":
            dist.all_gather(parts, mine)
        else:
            hp = [q.cpu() for q in parts]
            dist.all_gather(hp, mine.cpu())
            for q, h in zip(parts, hp):
                q.copy_(h)

    imt, jmt, km = (int(t) for t in a.grid.split("x"))
    t0 = time.perf_counter()
    k33 = a.recipe == "k33"
    p = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, isop_k33=k33)
    blk = solver.column_blocks(p.col_start(), p.tracer_state_len, 1)
    ci, cj = solver.column_coords(p.ind_i, p.ind_j, p.col_start(), 1)
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    kw = dict(device=local_rank, precond=solver.PRECOND_MULTILEVEL, restart=a.restart, ml_smooth=a.ml_smooth, rtol=a.rtol,
              max_iters=a.max_iters, rank=rank)
    n_global = p.flat_len
    nnz_global = p.nnz
    tracers_global = 1
    mode = "single GPU"
    fst = 0
    s = None
    if world > 1 or a.force_dist:
        # The reference's contiguous row-block partition (src/solve_ABdist.c:141-144), halo exchange + allreduce
        # through torch.distributed's RCCL communicator, multilevel hierarchy per rank with one ring of the neighbours' water columns where the cut is lateral (restricted additive Schwarz).
        try:
            if a.multi_gpu == "c4":
                # BASELINE configs[3]: 4 coupled tracers.  The reference stores them tracer-major (src/matrix.c:778-784), where
                # its contiguous row blocks (src/solve_ABdist.c:141-144) put the same-cell couplings off-rank in every row; the
                # rows are renumbered cell-major first, so the same rule cuts latitude bands of the whole
                # coupled system: couplings rank-local, halo = the band edge (--c4-order tracer keeps the tracer-major cut)
                tracers_global = 4
                p4 = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, isop_k33=k33, coupled_tracer_cnt=4)
                blk4 = solver.column_blocks(p4.col_start(), p4.tracer_state_len, 4)
                ci4, cj4 = solver.column_coords(p4.ind_i, p4.ind_j, p4.col_start(), 4)
                n_global, nnz_global = p4.flat_len, p4.nnz
                if a.c4_order == "cell":
                    loc, starts, _ = nd.cell_major_slice(p4.rowptr, p4.colind, p4.nzval, blk4, 4, world, rank, ci4, cj4)
                    cnt_loc = 4
                    mode = (f"configs[3]: 4-tracer coupled system ({n_global} rows) renumbered cell-major and split into {world} contiguous row blocks "
                            f"(a latitude band of all four tracers per GPU), halo alltoallv + allreduce over RCCL (torch.distributed), "
                            f"multilevel hierarchy per rank with one ring of the neighbours' water columns where the cut is lateral (restricted additive Schwarz)")
                else:
                    if 4 % world != 0 and world % 4 != 0:
                        raise ValueError(f"tracer-major configs[3] splits 4 tracers over 1, 2, 4, 8, ... ranks, not {world} (a rank would hold parts of two tracers)")
                    starts = nd.snap_partition(blk4, world)
                    loc = nd.local_slice(p4.rowptr, p4.colind, p4.nzval, blk4, starts, rank, ci4, cj4)
                    cnt_loc = max(1, 4 // world)
                    mode = (f"configs[3]: 4-tracer coupled system ({n_global} rows), tracer-major, split into {world} contiguous row blocks "
                            f"({'%d tracer(s)' % (4 // world) if world <= 4 else 'a latitude band of one tracer'} per GPU), halo alltoallv + "
                            f"allreduce over RCCL (torch.distributed), multilevel hierarchy per rank with one ring of the neighbours' water columns where the cut is lateral (restricted additive Schwarz)")
                del p4
            elif a.multi_gpu == "weak":
                tracers_global = world
                cnt_loc = 1
                loc, starts, n_global = nd.tracer_slice(p, rank, world)
                nnz_global = world * (p.nnz + (world - 1) * p.flat_len)
                mode = (f"weak scaling: {world}-tracer coupled system, one tracer ({p.flat_len} rows) per GPU, halo alltoallv "
                        f"({world - 1} x {p.flat_len} values per SpMV) + allreduce over RCCL (torch.distributed), multilevel hierarchy per rank with one ring of the neighbours' water columns where the cut is lateral (restricted additive Schwarz)")
            else:
                cnt_loc = 1
                starts = nd.snap_partition(blk, world)
                loc = nd.local_slice(p.rowptr, p.colind, p.nzval, blk, starts, rank, ci, cj)
                mode = f"strong scaling: rows split into {world} latitude bands, halo alltoallv + allreduce over RCCL (torch.distributed), multilevel hierarchy per rank with one ring of the neighbours' water columns where the cut is lateral (restricted additive Schwarz)"
            comm = None
            if a.comm == "rccl" and backend == "nccl":
                # the library's own communicator, checked before anything is built on it: a wrong answer or a call that
                # never returns sends EVERY rank to the torch.distributed callbacks (same decision on all ranks)
                good = 1
                try:
                    comm = nd.RcclComm()
                    nd.comm_self_test(comm, timeout=120.0)
                except Exception as exc:
                    print(f"({rank}) library RCCL communicator failed its self-test, using torch.distributed callbacks: {exc!r}", file=sys.stderr)
                    good = 0
                flag = torch.tensor([good], device="cuda")
                all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    comm = None
            if comm is None:
                comm = nd.TorchComm()
            mode += f"; collectives: {'library RCCL communicator (comm_rccl.hip)' if isinstance(comm, nd.RcclComm) else 'torch.distributed callbacks'}"
            s = nd.NkpDistSolver(loc, n_global, comm, coupled_tracer_cnt=cnt_loc, **kw)
            fst = loc["fst_row"]
        except Exception as exc:
            print(f"({rank}) distributed setup failed: {exc!r}", file=sys.stderr)
            s = None
        # a scaling run that silently measured N independent replicas would read as a distributed result: every rank
        # learns whether all of them built their solver, and the run ends non-zero if one did not
        ok = torch.tensor([1 if s is not None else 0], device="cuda")
        all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if dist.is_initialized():
                dist.destroy_process_group()
            raise SystemExit(f"({rank}) bench.py: the distributed solver could not be set up on every rank (see stderr); no line is printed")
    if world == 1 and not a.force_dist:
        s = solver.NkpSolver(p.rowptr, p.colind, p.nzval, blk, col_i=ci, col_j=cj, **kw)
    distributed = world > 1 or a.force_dist
    t_setup = time.perf_counter() - t0
    n = s.n                                               # rows this rank solves for

    # right-hand sides resident in HBM before the timed region
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1234 if distributed else 1234 + rank)
    nrhs = a.warmup + a.steps
    if distributed:                                       # every rank draws the same global rhs and keeps its slice
        B = torch.stack([torch.randn(n_global, dtype=torch.float64, device="cuda", generator=gen)[fst:fst + n] for _ in range(nrhs)])
    else:
        B = torch.randn((nrhs, n), dtype=torch.float64, device="cuda", generator=gen)
    X = torch.zeros((nrhs, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    if world > 1 or a.force_dist:
        a.no_cpu_baseline = a.no_cpu_baseline or world > 1     # the CPU leg is an N = 1 measurement

    infos = []
    first_solve_s = None
    for k in range(a.warmup):
        t1 = time.perf_counter()
        s.solve_device(B[k].data_ptr(), X[k].data_ptr())
        if first_solve_s is None:
            first_solve_s = time.perf_counter() - t1
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.warmup, nrhs):
        t1 = time.perf_counter()
        infos.append(s.solve_device(B[k].data_ptr(), X[k].data_ptr()))
        if first_solve_s is None:
            first_solve_s = time.perf_counter() - t1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # independent check of the last solution with torch's own SpMV (not the kernel under test)
    relres_check = None
    try:
        if distributed:
            sizes = [int(starts[r + 1] - starts[r]) for r in range(world)]
            pad = max(sizes)

            def gather_padded(v):                          # equal-size all_gather, then trim
                mine = torch.zeros(pad, dtype=torch.float64, device="cuda")
                mine[:v.numel()] = v
                parts = [torch.empty(pad, dtype=torch.float64, device="cuda") for _ in range(world)]
                all_gather(parts, mine)
                return torch.cat([parts[r][:sizes[r]] for r in range(world)])
            xg, bg = gather_padded(X[-1]), None
        else:
            xg, bg = X[-1], B[-1]
        if distributed:
            # every rank checks its own rows against the gathered solution; squared norms are summed over ranks
            crow = torch.from_numpy(loc["rowptr"].astype(np.int64)).cuda()
            ccol = torch.from_numpy(loc["colind"].astype(np.int64)).cuda()
            cval = torch.from_numpy(loc["val"]).cuda()
            At = torch.sparse_csr_tensor(crow, ccol, cval, size=(n, n_global))
            r = B[-1] - (At @ xg.unsqueeze(1)).squeeze(1)
            sq = torch.stack([torch.sum(r * r), torch.sum(B[-1] * B[-1])])
            all_reduce(sq)
            relres_check = float(torch.sqrt(sq[0] / sq[1]))
            del At, crow, ccol, cval, r
        else:
            crow = torch.from_numpy(p.rowptr.astype(np.int64)).cuda()
            ccol = torch.from_numpy(p.colind.astype(np.int64)).cuda()
            cval = torch.from_numpy(p.nzval).cuda()
            At = torch.sparse_csr_tensor(crow, ccol, cval, size=(n_global, n_global))
            r = bg - (At @ xg.unsqueeze(1)).squeeze(1)
            relres_check = float(torch.linalg.norm(r) / torch.linalg.norm(bg))
            del At, crow, ccol, cval
    except Exception as exc:
        print(f"({rank}) residual re-check skipped: {exc!r}", file=sys.stderr)

    # dominant-kernel roofline: the CSR SpMV, timed with HIP events on the solver's own stream
    spmv_ms = s.time_kernel(0, reps=200)
    spmv_bytes = s.get_int("spmv_bytes")
    achieved = spmv_bytes / spmv_ms / 1e6          # GB/s
    pre_ms = s.time_kernel(1, reps=50)
    it_ms = {f"j{j}": s.time_kernel(2, reps=10, arg=j) for j in (0, a.restart // 2, a.restart - 1)}

    # HBM bytes per SpMV launch from the PMC counters: collected in separate rocprofv3 --pmc passes
    # (FETCH_SIZE, WRITE_SIZE) and committed under profiles/; used only when it is this exact workload
    traffic = None
    try:
        rec = json.load(open(latest_profile("spmv_pmc.json")))
        if rec.get("n") == n_global and rec.get("nnz") == p.nnz and not distributed:
            traffic = rec["traffic_bytes_per_launch"]
    except Exception:
        pass
    # the kernels the solve actually spends its time in (HIP events on the solver's stream, compulsory bytes from the library)
    kernels = []
    if not distributed:
        for label, which, key, reps in (("csr_spmv_pipe_kernel<1, float> (smoother residual rows, fine level, one colour)", 3, "smoother_spmv_bytes", 100),
                                        ("colblock_apply_ldspack_kernel (water-column solves, fine level, one colour)", 4, "column_solve_bytes", 100),
                                        ("whole V-cycle (all levels, 88 launches)", 1, "cycle_bytes", 50)):
            ms = pre_ms if which == 1 else s.time_kernel(which, reps=reps)
            nbytes = s.get_int(key)
            if nbytes > 0 and ms > 0:
                kernels.append({"kernel": label, "bound": "hbm", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms,
                                "achieved": nbytes / ms / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / ms / 1e6 / HBM_PEAK_GBS})
    # HBM bytes per launch of those kernels from the PMC counters (separate rocprofv3 --pmc passes over tools/pmc_cycle.py,
    # committed under profiles/); attached only when it is this exact workload
    try:
        rec = json.load(open(latest_profile("cycle_pmc.json")))
        if rec.get("n") == n_global and rec.get("nnz") == p.nnz and not distributed:
            for k in kernels:
                for q in rec["kernels"]:
                    if k["kernel"].split("<")[0].split(" ")[0] in q["kernel"] and abs(q["algorithmic_bytes_per_launch"] - k["algorithmic_bytes_per_launch"]) <= 0.01 * k["algorithmic_bytes_per_launch"]:
                        k["traffic"] = q["traffic_bytes_per_launch"]
    except Exception:
        pass
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    iters = [i["iters"] for i in infos]
    tracer_text = f"{tracers_global}-tracer coupled" if (distributed and tracers_global > 1) else "single-tracer"
    deg = {320: "1", 100: "3", 640: "0.5", 1440: "0.25"}.get(imt, "?")
    recipe_text = "isop with the K33 term of the Redi tensor" if k33 else "isop, round-1 recipe (no K33 term)"
    out = {
        "metric": "precond_solve_throughput_1deg_ocean_jacobian",
        "value": (1 if distributed else world) * a.steps * n_global / dt,
        "unit": "unknowns/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if (distributed and a.multi_gpu in ("strong", "c4")) else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{a.grid} ({deg} degree x {km} level) {tracer_text} ocean Jacobian, "
                        f"adv={a.adv} hmix={a.hmix} ({recipe_text}), n={n_global}, nnz={nnz_global}; FGMRES({a.restart}) + multilevel water-column "
                        f"preconditioner V({a.ml_smooth},{a.ml_smooth}) x {s.get_int('precond_steps')} per iteration (defect correction), rtol={a.rtol:g}; "
                        f"one solve per step, rhs resident in HBM",
            "multi_gpu": mode,
        },
        "solve": {"iterations": iters, "relres": [i["relres"] for i in infos], "berr": [i["berr"] for i in infos],
                  "relres_checked_with_torch": relres_check, "setup_s": t_setup, "generate_s": t_gen,
                  # what the reference's factor-then-solve job waits for (src/solve_ABglobal.c:349-402): host arrays ready ->
                  # solver created (matrix upload, hierarchy, factors) -> first right-hand side solved, first-launch costs included
                  "first_solve_s": first_solve_s, "time_to_first_solution_s": t_setup + (first_solve_s or 0.0),
                  "create_s_in_library": s.get_int("create_us") / 1e6, "hierarchy_setup_s": s.get_int("ml_setup_us") / 1e6,
                  "levels_built_on_device": s.get_int("ml_levels_on_device"),
                  "levels": s.get_int("levels"), "precond_cycles_per_iteration": s.get_int("precond_steps"),
                  "device_MB": s.get_int("device_bytes") / 1e6,
                  "distributed": ({"halo_hidden_behind_interior_rows": s.get_int("dist_overlap"), "hierarchy_overlaps_neighbours": s.get_int("dist_ras"),
                                   "overlap_rows_rank0": s.get_int("dist_ras_rows")} if distributed else None),
                  "precond_apply_ms": pre_ms, "krylov_iteration_ms": it_ms},
        "roofline": {"kernel": "csr_spmv_pipe_kernel<0, double, false>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": spmv_bytes, "avg_launch_ms": spmv_ms,
                     # context only: the guide's measured float4-copy ceiling (MI355X_MICROARCH.md: 6.29 TB/s)
                     "measured_copy_ceiling": MEASURED_COPY_GBS, "frac_of_copy_ceiling": achieved / MEASURED_COPY_GBS,
                     "hbm_traffic_rate": (traffic / spmv_ms / 1e6) if traffic else None,
                     "kernels": kernels},
    }
    if not distributed and world == 1 and a.rhs_batch > 1:
        # the reference's RHS loop (src/solve_ABglobal.c:370-409) with R tracers per sweep of the matrix and the hierarchy
        # (nkp_solve_batch_device: one FGMRES recurrence per system, the operator and cycle applications shared); beside it the
        # round-2 way, R clones on R host threads
        R = a.rhs_batch
        gen2 = torch.Generator(device="cuda")
        gen2.manual_seed(4321)
        Bb = torch.randn((a.steps, R, n), dtype=torch.float64, device="cuda", generator=gen2)
        Xb = torch.zeros_like(Bb)
        torch.cuda.synchronize()
        s.solve_batch_device(Bb[0].data_ptr(), Xb[0].data_ptr(), R, n)              # first call allocates the extra work vectors
        torch.cuda.synchronize()
        tb = time.perf_counter()
        binfo = [s.solve_batch_device(Bb[k].data_ptr(), Xb[k].data_ptr(), R, n) for k in range(a.steps)]
        torch.cuda.synchronize()
        dtb = time.perf_counter() - tb
        out["rhs_batch"] = {"rhs_per_sweep": R, "solves": R * a.steps, "value": R * a.steps * n / dtb, "unit": "unknowns/s",
                            "ms_per_solve_effective": dtb / (R * a.steps) * 1e3, "speedup_vs_one_at_a_time": (dt / a.steps) / (dtb / (R * a.steps)),
                            "iterations": [i["iters"] for row in binfo for i in row], "max_relres": max(i["relres"] for row in binfo for i in row),
                            "how": "nkp_solve_batch_device: K interleaved right-hand sides per sweep (csrc/batch.hip); every column bit-identical to its single solve"}
        del Bb, Xb
        out["rhs_concurrent_clones"] = rhs_batch_throughput(torch, s, a.rhs_batch, a.steps, n)
    if not distributed and world == 1 and k33 and a.round1_steps > 0:
        # continuity with round 1: the same solver on the round-1 synthetic recipe (reported beside `value`, never as it)
        s.close()
        p1 = synth.generate(imt=imt, jmt=jmt, km=km, adv=a.adv, hmix=a.hmix, seed=0, isop_k33=False)
        s1 = solver.NkpSolver(p1.rowptr, p1.colind, p1.nzval, blk, col_i=ci, col_j=cj, **kw)
        B1 = torch.randn((a.round1_steps + 1, p1.flat_len), dtype=torch.float64, device="cuda", generator=gen)
        X1 = torch.zeros_like(B1)
        torch.cuda.synchronize()                       # B1 / X1 are filled on torch's stream, the solver reads them on its own
        s1.solve_device(B1[0].data_ptr(), X1[0].data_ptr())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        inf1 = [s1.solve_device(B1[k].data_ptr(), X1[k].data_ptr()) for k in range(1, a.round1_steps + 1)]
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t1
        out["round1_recipe"] = {"workload": "same grid and options, isopycnal cross terms without the K33 term (round 1's bench matrix)",
                                "ms_per_step": dt1 / a.round1_steps * 1e3, "value": a.round1_steps * p1.flat_len / dt1, "unit": "unknowns/s",
                                "iterations": [i["iters"] for i in inf1], "relres": [i["relres"] for i in inf1],
                                "round1_ms_per_step": 2861.96}
        s1.close()
        del p1
    if not a.no_cpu_baseline:
        # measured, not extrapolated: ONE complete solve of this very workload with the same algorithm on all host cores
        cs = cpu_complete_solve(p, blk, ci, cj, a.rtol, a.restart)
        out["cpu_baseline"] = {"value": cs["unknowns_per_s"], "unit": "unknowns/s", "cores": cs["cores"], "kind": "port",
                               "sample": f"one complete solve of the bench workload itself ({a.grid}, n = {cs['n']}, nnz = {cs['nnz']}) to rtol {a.rtol:g} with the same "
                                         f"algorithm (FGMRES({a.restart}) + multilevel water-column V(3,3) cycle, {cs['levels']} levels) in its C / OpenMP restatement "
                                         f"oracle/ml_oracle.c on {cs['cores']} host threads: {cs['solve_s']:.2f} s, {cs['iterations']} iterations, relres {cs['relres']:.1e} "
                                         f"(hierarchy setup {cs['setup_s']:.1f} s on the host not counted, like the GPU's setup)",
                               "complete_solve": cs}
    print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
