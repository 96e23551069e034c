"""Row-distributed flavour (one process per GPU): partition helpers, a torch.distributed-backed
implementation of the four collectives the C library needs (include/nkp.h `nkp_comm_ops`), and the
NkpDistSolver wrapper around nkp_create_dist.

Mirrors the reference's distributed executable (src/solve_ABdist.c): contiguous row blocks
(:141-144, snapped here to water-column boundaries so that a column never straddles ranks),
local rowptr rebased to 0 with GLOBAL column indices (:170-175, 188-225).

backend "nccl" (= RCCL on ROCm): the collectives act directly on the library's device buffers.
backend "gloo": the same calls staged through host memory (CPU tests; 2 ranks sharing one GPU).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import solver as _solver


def snap_partition(blk_start, nranks):
    """Row offsets starts[nranks+1]: the reference's n/P split, each cut moved to the nearest
    water-column boundary."""
    blk_start = np.asarray(blk_start, np.int64)
    n = int(blk_start[-1])
    starts = [0]
    for r in range(1, nranks):
        target = r * (n // nranks)
        q = int(np.searchsorted(blk_start, target))
        cand = [blk_start[max(q - 1, 0)], blk_start[min(q, blk_start.size - 1)]]
        cut = int(min(cand, key=lambda c: abs(int(c) - target)))
        starts.append(max(cut, starts[-1]))
    starts.append(n)
    return np.asarray(starts, np.int64)


def local_slice(rowptr, colind, val, blk_start, starts, rank, col_i=None, col_j=None):
    """This rank's rows: rebased rowptr, GLOBAL colind, values, local block offsets (+ coords)."""
    f, e = int(starts[rank]), int(starts[rank + 1])
    rp = np.asarray(rowptr[f:e + 1], np.int64)
    lo, hi = int(rp[0]), int(rp[-1])
    blk_start = np.asarray(blk_start, np.int64)
    b0, b1 = int(np.searchsorted(blk_start, f)), int(np.searchsorted(blk_start, e))
    if blk_start[b0] != f or blk_start[b1] != e:
        raise ValueError("partition cuts a water column")
    out = dict(rowptr=(rp - lo).astype(np.int32), colind=np.ascontiguousarray(colind[lo:hi], np.int32),
               val=np.ascontiguousarray(val[lo:hi], np.float64), blk_start=(blk_start[b0:b1 + 1] - f).astype(np.int32),
               fst_row=f, m_loc=e - f)
    if col_i is not None:
        out["col_i"] = np.ascontiguousarray(col_i[b0:b1], np.int32)
        out["col_j"] = np.ascontiguousarray(col_j[b0:b1], np.int32)
    return out


def tracer_slice(p1, rank, nranks):
    """One tracer per rank (weak scaling over coupled tracers, BASELINE config "1 degree x 4 tracers"): rows of
    tracer `rank` of the `nranks`-tracer coupled problem on the single-tracer synthetic problem p1.  Rows are
    tracer-major (reference src/matrix.c:778-784), so this IS the reference's contiguous row-block partition
    (src/solve_ABdist.c:141-144) with nranks == coupled_tracer_cnt.  Returns (loc, starts, n_global)."""
    from . import synth
    tsl = p1.tracer_state_len
    rp, ci, v = synth.tracer_rows(p1, rank, nranks)
    blk = _solver.column_blocks(p1.col_start(), tsl, 1)
    col_i, col_j = _solver.column_coords(p1.ind_i, p1.ind_j, p1.col_start(), 1)
    loc = dict(rowptr=rp, colind=ci, val=v, blk_start=blk, fst_row=rank * tsl, m_loc=tsl, col_i=col_i, col_j=col_j)
    return loc, np.arange(nranks + 1, dtype=np.int64) * tsl, nranks * tsl


def cell_major_slice(rowptr, colind, val, blk_start, coupled_tracer_cnt, nranks, rank, col_i=None, col_j=None):
    """This rank's rows of a coupled system in CELL-MAJOR order: the tracer-major system
    (reference src/matrix.c:778-784) is renumbered so that the columns of all tracers at one water-column position follow
    each other, then cut by the reference's contiguous row-block rule (src/solve_ABdist.c:141-144).  The blocks are latitude
    bands of the whole coupled system: the same-cell couplings stay on the rank, the halo is the band edge.
    Returns (loc, starts, perm): loc as local_slice gives it (plus col_t), perm = new row -> old row (b_new = b_old[perm])."""
    perm, inv, blk_new, col_t, col_src = _solver.cell_major_order(blk_start, coupled_tracer_cnt)
    # cuts between cells, never between the tracers of one cell
    starts = snap_partition(blk_new[::coupled_tracer_cnt], nranks)
    f, e = int(starts[rank]), int(starts[rank + 1])
    rp, ci, v = _solver.permuted_rows(rowptr, colind, val, perm, inv, f, e)
    b0, b1 = int(np.searchsorted(blk_new, f)), int(np.searchsorted(blk_new, e))
    loc = dict(rowptr=rp, colind=ci, val=v, blk_start=(blk_new[b0:b1 + 1].astype(np.int64) - f).astype(np.int32), fst_row=f, m_loc=e - f,
               col_t=np.ascontiguousarray(col_t[b0:b1], np.int32))
    if col_i is not None:
        loc["col_i"] = np.ascontiguousarray(np.asarray(col_i)[col_src[b0:b1]], np.int32)
        loc["col_j"] = np.ascontiguousarray(np.asarray(col_j)[col_src[b0:b1]], np.int32)
    return loc, starts, perm


def plan_host(rowptr_loc, colind_glob, starts, rank):
    """nkp_dist_plan_host: remapped columns, needed off-rank rows, per-owner counts (no GPU needed)."""
    lib = _solver.load_library()
    rp = np.ascontiguousarray(rowptr_loc, np.int32)
    ci = np.ascontiguousarray(colind_glob, np.int32)
    st = np.ascontiguousarray(starts, np.int64)
    P = st.size - 1
    ext = np.empty(max(ci.size, 1), np.int32)
    halo = np.empty(max(ci.size, 1), np.int32)
    need = np.zeros(P, np.int32)
    nh = C.c_int64()
    rc = lib.nkp_dist_plan_host(rp.size - 1, ci.size, rp.ctypes.data_as(C.POINTER(C.c_int32)), ci.ctypes.data_as(C.POINTER(C.c_int32)),
                                rank, P, st.ctypes.data_as(C.POINTER(C.c_int64)), ext.ctypes.data_as(C.POINTER(C.c_int32)),
                                halo.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nh), need.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise _solver.NkpError(rc, lib.nkp_last_error().decode())
    return ext[:ci.size], halo[:nh.value].copy(), need


def overlap_plan_host(loc, n_global, comm, coupled_tracer_cnt=1):
    """nkp_dist_overlap_plan_host: everything nkp_create_dist decides on the host (halo of the SpMV, overlap of the
    hierarchy), as a dict of numpy arrays.  Collective; needs no GPU (the callbacks used are the host ones)."""
    lib = _solver.load_library()
    opt = _solver.default_options()
    rp, ci, v, bs = loc["rowptr"], loc["colind"], loc["val"], loc["blk_start"]
    keep = [np.ascontiguousarray(loc[k], np.int32) for k in ("col_i", "col_j", "col_t") if loc.get(k) is not None]
    if loc.get("col_i") is not None:
        opt.col_i, opt.col_j = _solver._p(keep[0], C.c_int32), _solver._p(keep[1], C.c_int32)
    if loc.get("col_t") is not None:
        opt.col_t = _solver._p(keep[-1], C.c_int32)
    h = C.c_void_p()
    lib.nkp_dist_overlap_plan_host.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.c_int, C.c_void_p]
    lib.nkp_dist_plan_size.restype = C.c_int64
    lib.nkp_dist_plan_size.argtypes = [C.c_void_p, C.c_char_p]
    lib.nkp_dist_plan_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    lib.nkp_dist_plan_free.argtypes = [C.c_void_p]
    rc = lib.nkp_dist_overlap_plan_host(C.byref(h), C.cast(C.byref(opt), C.c_void_p), int(n_global), int(loc["fst_row"]), int(loc["m_loc"]), int(ci.size),
                                        _solver._p(rp, C.c_int32), _solver._p(ci, C.c_int32), _solver._p(v, C.c_double), _solver._p(bs, C.c_int32),
                                        int(bs.size - 1), int(coupled_tracer_cnt), C.cast(C.byref(comm.ops), C.c_void_p))
    if rc != 0:
        raise _solver.NkpError(rc, lib.nkp_last_error().decode() + (" | comm: " + "; ".join(comm.errors) if comm.errors else ""))
    out = dict(ras=int(lib.nkp_dist_plan_size(h, b"ras")))
    for name in ("colind_ext", "halo_rows", "send_rows", "need", "give", "rowptr", "colind", "val", "blk_start", "col_i", "col_j", "col_t", "sel_hpos"):
        cnt = int(lib.nkp_dist_plan_size(h, name.encode()))
        arr = np.empty(max(cnt, 0), np.float64 if name == "val" else np.int32)
        if cnt > 0:
            lib.nkp_dist_plan_copy(h, name.encode(), arr.ctypes.data_as(C.c_void_p))
        out[name] = arr
    lib.nkp_dist_plan_free(h)
    return out


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=3)


class TorchComm:
    """nkp_comm_ops on top of an initialised torch.distributed process group."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
        self.device_native = dist.get_backend() == "nccl"
        self.errors = []
        self._views = {}
        self._streams = {}
        self._fns = (_solver._ALLREDUCE_FN(self._allreduce), _solver._ALLTOALLV_FN(self._alltoallv),
                     _solver._ALLTOALLV_I32_FN(self._alltoallv_i32_host), _solver._ALLGATHER_I64_FN(self._allgather_i64_host))
        self.ops = _solver.NkpCommOps(None, self.rank, self.nranks, *self._fns)

    # ---- helpers
    def _dev(self, ptr, n):
        # the library's device buffers live as long as the solver: wrap each (address, length) once
        key = (int(ptr), int(n))
        t = self._views.get(key)
        if t is None:
            t = self.torch.as_tensor(_DevArray(ptr, n, "<f8"), device="cuda")
            if len(self._views) < 4096:
                self._views[key] = t
        return t

    def _on(self, stream):
        """The collective is ordered on the HIP stream the library names (its own non-blocking stream unless the caller
        installed another): torch enqueues on whatever stream is current, so make that stream current for the call."""
        ptr = int(stream) if stream else 0
        st = self._streams.get(ptr)
        if st is None:
            st = self.torch.cuda.ExternalStream(ptr) if ptr else self.torch.cuda.default_stream()
            self._streams[ptr] = st
        return self.torch.cuda.stream(st)

    def _exchange_host(self, send, scnt, recv, rcnt):
        """alltoallv of 1-D CPU tensors with isend/irecv (gloo has no all_to_all)."""
        dist = self.dist
        so = np.concatenate([[0], np.cumsum(scnt)]).astype(np.int64)
        ro = np.concatenate([[0], np.cumsum(rcnt)]).astype(np.int64)
        reqs = []
        for p in range(self.nranks):
            if p == self.rank:
                if scnt[p]:
                    recv[ro[p]:ro[p + 1]] = send[so[p]:so[p + 1]]
                continue
            if rcnt[p]:
                reqs.append(dist.irecv(recv[ro[p]:ro[p + 1]], src=p))
            if scnt[p]:
                reqs.append(dist.isend(send[so[p]:so[p + 1]].contiguous(), dst=p))
        for r in reqs:
            r.wait()

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as exc:                      # never let an exception cross the C boundary
            self.errors.append(repr(exc))
            return 1

    # ---- the four collectives
    def _allreduce(self, ctx, dev_buf, count, op, stream):
        def run():
            with self._on(stream):
                t = self._dev(dev_buf, count)
                rop = self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM
                if self.device_native:
                    self.dist.all_reduce(t, op=rop)
                else:
                    h = t.cpu()                      # synchronises the stream: the library's kernels have finished
                    self.dist.all_reduce(h, op=rop)
                    t.copy_(h)
                    self.torch.cuda.current_stream().synchronize()
        return self._guard(run)

    def _alltoallv(self, ctx, dev_send, scnt, dev_recv, rcnt, stream):
        def run():
            sc = [int(scnt[p]) for p in range(self.nranks)]
            rc = [int(rcnt[p]) for p in range(self.nranks)]
            ns, nr = sum(sc), sum(rc)
            with self._on(stream):
                send = self._dev(dev_send, ns) if ns else self.torch.empty(0, dtype=self.torch.float64, device="cuda")
                recv = self._dev(dev_recv, nr) if nr else self.torch.empty(0, dtype=self.torch.float64, device="cuda")
                if self.device_native:
                    self.dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc)
                else:
                    hs, hr = send.cpu(), self.torch.empty(nr, dtype=self.torch.float64)
                    self._exchange_host(hs, sc, hr, rc)
                    if nr:
                        recv.copy_(hr)
                        self.torch.cuda.current_stream().synchronize()
        return self._guard(run)

    def _alltoallv_i32_host(self, ctx, send, scnt, recv, rcnt):
        def run():
            torch = self.torch
            sc = [int(scnt[p]) for p in range(self.nranks)]
            rc = [int(rcnt[p]) for p in range(self.nranks)]
            ns, nr = sum(sc), sum(rc)
            hs = torch.from_numpy(np.ctypeslib.as_array(send, (max(ns, 1),))[:ns].copy())
            hr = torch.empty(nr, dtype=torch.int32)
            if self.device_native:
                ds, dr = hs.cuda(), torch.empty(nr, dtype=torch.int32, device="cuda")
                self.dist.all_to_all_single(dr, ds, output_split_sizes=rc, input_split_sizes=sc)
                hr = dr.cpu()
            else:
                self._exchange_host(hs, sc, hr, rc)
            if nr:
                np.ctypeslib.as_array(recv, (nr,))[:] = hr.numpy()
        return self._guard(run)

    def _allgather_i64_host(self, ctx, mine, out):
        def run():
            torch = self.torch
            dev = "cuda" if self.device_native else "cpu"
            t = torch.tensor([int(mine)], dtype=torch.int64, device=dev)
            parts = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(self.nranks)]
            self.dist.all_gather(parts, t)
            for p in range(self.nranks):
                out[p] = int(parts[p].item())
        return self._guard(run)


class RcclComm:
    """nkp_comm_ops backed by the library's own RCCL communicator (csrc/comm_rccl.hip): ncclAllReduce and grouped
    ncclSend / ncclRecv enqueued on the solver's stream straight from C -- no Python in the per-iteration collectives.
    torch.distributed is only used once, to hand rank 0's RCCL unique id to the other ranks."""

    def __init__(self):
        import torch.distributed as dist
        lib = _solver.load_library()
        self._lib = lib
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()
        self.errors = []
        ident = (C.c_ubyte * 128)()
        made = self.rank != 0 or lib.nkp_comm_unique_id(ident) == 0
        box = [bytes(ident) if made else None]        # rank 0's failure reaches every rank instead of leaving them waiting
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise _solver.NkpError(-5, "nkp_comm_unique_id failed on rank 0")
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        self.ops = _solver.NkpCommOps()
        lib.nkp_comm_rccl_init.argtypes = [C.POINTER(_solver.NkpCommOps), C.c_void_p, C.c_int, C.c_int]
        rc = lib.nkp_comm_rccl_init(C.byref(self.ops), C.cast(ident, C.c_void_p), self.rank, self.nranks)
        if rc != 0:
            raise _solver.NkpError(rc, "nkp_comm_rccl_init failed")

    def close(self):
        if self.ops.ctx:
            self._lib.nkp_comm_rccl_free.argtypes = [C.POINTER(_solver.NkpCommOps)]
            self._lib.nkp_comm_rccl_free(C.byref(self.ops))


def comm_self_test(comm, timeout=120.0):
    """Pre-flight of a transport before a solver is built on it: one sum and one max allreduce and one personalised
    exchange with rank-dependent counts, checked against the values every rank must end with.  The calls run in a
    helper thread so that a transport that never returns is reported (TimeoutError) instead of hanging the job.
    Collective: every rank calls it."""
    import threading
    import torch
    ops, rank, nranks = comm.ops, comm.ops.rank, comm.ops.nranks
    device = torch.cuda.current_device()
    out = {}

    def cnt(a, b):                                  # symmetric, so what a sends b is what b expects from a
        return (a + b) % 3 + 1

    def run():
        try:
            torch.cuda.set_device(device)
            st = torch.cuda.Stream()
            sp = C.c_void_p(st.cuda_stream)
            buf = torch.full((4,), float(rank + 1), dtype=torch.float64, device="cuda")
            st.wait_stream(torch.cuda.current_stream())
            if ops.allreduce(ops.ctx, C.c_void_p(buf.data_ptr()), 4, 0, sp):
                raise RuntimeError("allreduce(sum) returned an error")
            st.synchronize()
            want = nranks * (nranks + 1) / 2.0
            if not bool((buf == want).all()):
                raise RuntimeError(f"allreduce(sum) gave {buf.tolist()}, expected {want}")
            buf.fill_(float(rank))
            torch.cuda.current_stream().synchronize()
            if ops.allreduce(ops.ctx, C.c_void_p(buf.data_ptr()), 4, 1, sp):
                raise RuntimeError("allreduce(max) returned an error")
            st.synchronize()
            if not bool((buf == float(nranks - 1)).all()):
                raise RuntimeError(f"allreduce(max) gave {buf.tolist()}")
            sc = (C.c_int * nranks)(*[cnt(rank, p) for p in range(nranks)])
            send = torch.cat([torch.full((cnt(rank, p),), 100.0 * rank + p, dtype=torch.float64) for p in range(nranks)]).cuda()
            recv = torch.full((int(send.numel()),), -1.0, dtype=torch.float64, device="cuda")
            torch.cuda.current_stream().synchronize()
            if ops.alltoallv(ops.ctx, C.c_void_p(send.data_ptr()), sc, C.c_void_p(recv.data_ptr()), sc, sp):
                raise RuntimeError("alltoallv returned an error")
            st.synchronize()
            expect = torch.cat([torch.full((cnt(rank, p),), 100.0 * p + rank, dtype=torch.float64) for p in range(nranks)])
            if not bool((recv.cpu() == expect).all()):
                raise RuntimeError("alltoallv delivered the wrong values")
            out["ok"] = True
        except Exception as exc:
            out["error"] = exc

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        raise TimeoutError(f"transport self-test did not return within {timeout:.0f} s")
    if "error" in out:
        raise out["error"]
    return True


class NkpDistSolver(_solver.NkpSolver):
    """nkp_create_dist: this rank's row block in, a solver for the LOCAL slices of b / x out."""

    def __init__(self, loc, n_global, comm, coupled_tracer_cnt=1, **options):
        lib = _solver.load_library()
        self._comm = comm                      # keeps the callbacks alive
        opt = _solver.default_options(**options)
        rp, ci, v, bs = loc["rowptr"], loc["colind"], loc["val"], loc["blk_start"]
        self._keep = (rp, ci, v, bs, loc.get("col_i"), loc.get("col_j"), loc.get("col_t"))
        if loc.get("col_i") is not None:
            opt.col_i, opt.col_j = _solver._p(loc["col_i"], C.c_int32), _solver._p(loc["col_j"], C.c_int32)
        if loc.get("col_t") is not None:
            opt.col_t = _solver._p(loc["col_t"], C.c_int32)
        h = C.c_void_p()
        rc = lib.nkp_create_dist(C.byref(h), C.byref(opt), int(n_global), int(loc["fst_row"]), int(loc["m_loc"]), int(ci.size),
                                 _solver._p(rp, C.c_int32), _solver._p(ci, C.c_int32), _solver._p(v, C.c_double),
                                 _solver._p(bs, C.c_int32), int(bs.size - 1), coupled_tracer_cnt, C.byref(comm.ops))
        if rc != 0:
            raise _solver.NkpError(rc, lib.nkp_last_error().decode() + (" | comm: " + "; ".join(comm.errors) if comm.errors else ""))
        self._lib, self._h, self.n, self.nnz, self.options = lib, h, int(loc["m_loc"]), int(ci.size), opt
