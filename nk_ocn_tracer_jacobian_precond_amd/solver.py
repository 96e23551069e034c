"""ctypes binding of the C ABI in include/nkp.h (libnkp_hip.so) and of the host library.

This is the Python host-side mirror of the boundary the reference's executables cross when they
call SuperLU_DIST (reference src/solve_ABglobal.c:327-424): create ("factor") once, solve per
right-hand side with B overwritten by X, destroy.  There is no CPU fallback: if the HIP library
is missing or no GPU is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libnkp_hip.so")
HOST_LIB_PATH = os.path.join(_HERE, "host", "libnkp_host.so")

PRECOND_NONE, PRECOND_COLUMN_JACOBI, PRECOND_MULTILEVEL = 0, 1, 3
KRYLOV_FGMRES, KRYLOV_BICGSTAB = 0, 1
NKP_OK, NKP_NOT_CONVERGED, NKP_BREAKDOWN, NKP_OK_BERR = 0, 1, 2, 3

# every symbol include/nkp.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "nkp_default_options", "nkp_device_count", "nkp_create", "nkp_solve", "nkp_solve_device",
    "nkp_spmv", "nkp_spmv_device", "nkp_precond_apply", "nkp_multi_dot", "nkp_time_kernel",
    "nkp_get_int", "nkp_set_stream", "nkp_destroy", "nkp_last_error", "nkp_comm_unique_id",
    "nkp_comm_rccl_init", "nkp_comm_rccl_free", "nkp_create_dist", "nkp_dist_plan_host", "nkp_set_device",
    "nkp_gather_root", "nkp_clone", "nkp_ml_plan_host", "nkp_comm_file_init", "nkp_comm_file_free",
    "nkp_create64", "nkp_cell_major_order", "nkp_permuted_rows", "nkp_dist_overlap_plan_host", "nkp_dist_plan_size",
    "nkp_dist_plan_copy", "nkp_dist_plan_free", "nkp_ml_level_array", "nkp_default_tuning", "nkp_solve_batch_device",
]

_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
_ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int), C.c_void_p)
_ALLTOALLV_I32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int32), C.POINTER(C.c_int))
_ALLGATHER_I64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64))


class NkpCommOps(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int), ("allreduce", _ALLREDUCE_FN),
                ("alltoallv", _ALLTOALLV_FN), ("alltoallv_i32_host", _ALLTOALLV_I32_FN),
                ("allgather_i64_host", _ALLGATHER_I64_FN)]


class NkpTuning(C.Structure):
    """nkp_tuning (include/nkp.h): the knobs behind the defaults; NKP_* environment variables set the same fields when no struct is passed."""
    _fields_ = [
        ("struct_size", C.c_int), ("ml_split", C.c_int), ("ml_pocket", C.c_int), ("ml_big_from", C.c_int), ("ml_coarsest_rows", C.c_int),
        ("ml_dense_max", C.c_int), ("ml_theta", C.c_double), ("ml_tau", C.c_double), ("ml_device_min", C.c_int64),
        ("ml_smooth_coarse", C.c_int), ("ml_coarse_from", C.c_int), ("ml_gamma_from", C.c_int), ("ml_gamma_to", C.c_int), ("ml_f32", C.c_int),
        ("ml_host_inverse", C.c_int), ("ml_fused", C.c_int), ("ml_fused_max_cols", C.c_int), ("ml_wave_fused", C.c_int), ("ml_coarsest_sweeps", C.c_int),
        ("ml_tail_rows", C.c_int64), ("ml_omega", C.c_double),
        ("col_ldsres", C.c_int), ("col_stream", C.c_int), ("col_stream_min", C.c_int), ("col_stream_gw", C.c_int), ("col_wave_max", C.c_int),
        ("col_w3", C.c_int), ("col_group", C.c_int), ("col_pipe_min", C.c_int), ("col_ldsres_early", C.c_int), ("col_ldsres_packed", C.c_int),
        ("spmv_variant", C.c_int), ("spmv_compress", C.c_int), ("spmv_pipe_min", C.c_int), ("spmv_run", C.c_int), ("spmv_wgs", C.c_int),
        ("rhs_batch", C.c_int), ("precond_steps", C.c_int), ("equil", C.c_int), ("dist_overlap", C.c_int), ("dist_ras", C.c_int), ("force_dist", C.c_int),
        ("setup_threads", C.c_int), ("plan_times", C.c_int), ("ml_drop_intertracer", C.c_int), ("dist_one_reduce", C.c_int), ("ml_huge_from", C.c_int), ("col_ldsres_min", C.c_int), ("col_sort_groups", C.c_int), ("batch_spmv_rows", C.c_int),
    ]


class NkpOptions(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int), ("precond", C.c_int), ("krylov", C.c_int), ("restart", C.c_int),
        ("max_iters", C.c_int), ("rtol", C.c_double), ("atol", C.c_double), ("device", C.c_int),
        ("verbose", C.c_int), ("rank", C.c_int), ("reorth", C.c_int), ("ml_levels", C.c_int),
        ("ml_smooth", C.c_int), ("basis_f32", C.c_int), ("precond_steps", C.c_int), ("equil", C.c_int), ("reserved", C.c_int * 4),
        ("col_i", C.POINTER(C.c_int32)), ("col_j", C.POINTER(C.c_int32)), ("col_t", C.POINTER(C.c_int32)),
        ("tuning", C.POINTER(NkpTuning)),
    ]


class NkpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"nkp error {code}: {message}")
        self.code = code


_lib = None


def load_library(path=None):
    """Load libnkp_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or HIP_LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the solve path)")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    i32p, f64p, vp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_void_p
    lib.nkp_default_options.argtypes = [C.POINTER(NkpOptions)]
    lib.nkp_device_count.restype = C.c_int
    lib.nkp_create.argtypes = [C.POINTER(vp), C.POINTER(NkpOptions), C.c_int64, C.c_int64, i32p, i32p, f64p, i32p, C.c_int64, C.c_int]
    lib.nkp_create64.argtypes = [C.POINTER(vp), C.POINTER(NkpOptions), C.c_int64, C.POINTER(C.c_int64), i32p, f64p, i32p, C.c_int64, C.c_int]
    lib.nkp_solve.argtypes = [vp, f64p, C.c_int, C.c_int64, f64p, C.POINTER(C.c_int), f64p]
    lib.nkp_solve_device.argtypes = [vp, vp, vp, C.c_int, f64p, C.POINTER(C.c_int), f64p]
    lib.nkp_solve_batch_device.argtypes = [vp, C.c_int, vp, vp, C.c_int64, f64p, C.POINTER(C.c_int), f64p]
    lib.nkp_spmv.argtypes = [vp, f64p, f64p]
    lib.nkp_spmv_device.argtypes = [vp, vp, vp]
    lib.nkp_precond_apply.argtypes = [vp, f64p, f64p]
    lib.nkp_multi_dot.argtypes = [vp, f64p, C.c_int64, C.c_int, f64p, f64p]
    lib.nkp_time_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, f64p]
    lib.nkp_get_int.argtypes = [vp, C.c_char_p]
    lib.nkp_get_int.restype = C.c_int64
    lib.nkp_ml_level_array.argtypes = [vp, C.c_int, C.c_char_p, vp, C.c_int64]
    lib.nkp_ml_level_array.restype = C.c_int64
    lib.nkp_set_stream.argtypes = [vp, vp]
    lib.nkp_destroy.argtypes = [vp]
    lib.nkp_destroy.restype = None
    lib.nkp_last_error.restype = C.c_char_p
    lib.nkp_comm_unique_id.argtypes = [vp]
    lib.nkp_comm_rccl_init.argtypes = [C.POINTER(NkpCommOps), vp, C.c_int, C.c_int]
    lib.nkp_comm_rccl_free.argtypes = [C.POINTER(NkpCommOps)]
    lib.nkp_comm_rccl_free.restype = None
    lib.nkp_create_dist.argtypes = [C.POINTER(vp), C.POINTER(NkpOptions), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                    i32p, i32p, f64p, i32p, C.c_int64, C.c_int, C.POINTER(NkpCommOps)]
    lib.nkp_set_device.argtypes = [C.c_int]
    lib.nkp_gather_root.argtypes = [vp, f64p, f64p]
    lib.nkp_clone.argtypes = [vp, C.POINTER(vp)]
    lib.nkp_dist_plan_host.argtypes = [C.c_int64, C.c_int64, i32p, i32p, C.c_int, C.c_int, C.POINTER(C.c_int64), i32p, i32p,
                                       C.POINTER(C.c_int64), i32p]
    lib.nkp_ml_plan_host.argtypes = [C.c_int64, i32p, i32p, f64p, i32p, C.c_int64, i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int64,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int64), i32p, i32p]
    if path == HIP_LIB_PATH:
        _lib = lib
    return lib


def ml_plan_host(rowptr, colind, val, blk_start, col_i=None, col_j=None, coupled_tracer_cnt=1, max_levels=0, coarsest_rows=8000):
    """nkp_ml_plan_host: the coarse cells of every level of the multilevel preconditioner (host only, no GPU needed).
    Returns (rows per level, [cmap of level l -> l+1], [column block of every row of level l+1])."""
    lib = load_library()
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colind = np.ascontiguousarray(colind, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    blk_start = np.ascontiguousarray(blk_start, np.int32)
    n = rowptr.size - 1
    ip = lambda a: None if a is None else _p(np.ascontiguousarray(a, np.int32), C.c_int32)
    ci = None if col_i is None else np.ascontiguousarray(col_i, np.int32)
    cj = None if col_j is None else np.ascontiguousarray(col_j, np.int32)
    cap = 2 * n + 64
    cmap, colof = np.empty(cap, np.int32), np.empty(cap, np.int32)
    rows = np.zeros(64, np.int64)
    nlev = C.c_int()
    rc = lib.nkp_ml_plan_host(n, _p(rowptr, C.c_int32), _p(colind, C.c_int32), _p(val, C.c_double), _p(blk_start, C.c_int32), blk_start.size - 1,
                              None if ci is None else _p(ci, C.c_int32), None if cj is None else _p(cj, C.c_int32), coupled_tracer_cnt, max_levels,
                              coarsest_rows, cap, C.byref(nlev), _p(rows, C.c_int64), _p(cmap, C.c_int32), _p(colof, C.c_int32))
    if rc != 0:
        raise NkpError(rc, "nkp_ml_plan_host failed")
    rows = rows[:nlev.value]
    cmaps, colofs, qc, qo = [], [], 0, 0
    for l in range(nlev.value - 1):
        cmaps.append(cmap[qc:qc + rows[l]].copy())
        colofs.append(colof[qo:qo + rows[l + 1]].copy())
        qc += rows[l]
        qo += rows[l + 1]
    return rows, cmaps, colofs


def default_options(**overrides):
    lib = load_library()
    o = NkpOptions()
    lib.nkp_default_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(f"nkp_options has no field {k}")
        setattr(o, k, v)
    return o


def default_tuning(**overrides):
    """nkp_default_tuning (defaults + NKP_* environment) with keyword overrides; pass the result as NkpSolver(..., tuning=t)."""
    lib = load_library()
    t = NkpTuning()
    lib.nkp_default_tuning.argtypes = [C.POINTER(NkpTuning)]
    lib.nkp_default_tuning(C.byref(t))
    for k, v in overrides.items():
        if not hasattr(t, k):
            raise AttributeError(f"nkp_tuning has no field {k}")
        setattr(t, k, v)
    return t


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def cell_major_order(blk_start, coupled_tracer_cnt):
    """nkp_cell_major_order (host only): the cell-major ordering of a tracer-major coupled system.  Returns
    (perm new row -> old row, inv old row -> new row, blk_start_new, col_t, col_src)."""
    lib = load_library()
    blk = np.ascontiguousarray(blk_start, np.int32)
    nblk, n = blk.size - 1, int(blk[-1])
    perm = np.empty(max(n, 1), np.int32)
    blk_new = np.empty(nblk + 1, np.int32)
    col_t = np.empty(max(nblk, 1), np.int32)
    col_src = np.empty(max(nblk, 1), np.int32)
    lib.nkp_cell_major_order.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.c_int] + [C.POINTER(C.c_int32)] * 4
    rc = lib.nkp_cell_major_order(nblk, _p(blk, C.c_int32), int(coupled_tracer_cnt), _p(perm, C.c_int32), _p(blk_new, C.c_int32),
                                  _p(col_t, C.c_int32), _p(col_src, C.c_int32))
    if rc != 0:
        raise NkpError(rc, lib.nkp_last_error().decode())
    perm = perm[:n]
    inv = np.empty(n, np.int32)
    inv[perm] = np.arange(n, dtype=np.int32)
    return perm, inv, blk_new, col_t[:nblk], col_src[:nblk]


def permuted_rows(rowptr, colind, val, perm, inv, r0, r1):
    """nkp_permuted_rows (host only): rows [r0, r1) of P A P^T as (rowptr rebased to 0, colind in the new numbering, val)."""
    lib = load_library()
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    colind = np.ascontiguousarray(colind, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    perm = np.ascontiguousarray(perm, np.int32)
    inv = np.ascontiguousarray(inv, np.int32)
    old = perm[r0:r1]
    nloc = int((rowptr[old.astype(np.int64) + 1].astype(np.int64) - rowptr[old]).sum())
    rp = np.empty(r1 - r0 + 1, np.int32)
    ci = np.empty(max(nloc, 1), np.int32)
    v = np.empty(max(nloc, 1), np.float64)
    lib.nkp_permuted_rows.argtypes = [C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    rc = lib.nkp_permuted_rows(rowptr.size - 1, _p(rowptr, C.c_int32), _p(colind, C.c_int32), _p(val, C.c_double), _p(perm, C.c_int32),
                               _p(inv, C.c_int32), int(r0), int(r1), _p(rp, C.c_int32), _p(ci, C.c_int32), _p(v, C.c_double))
    if rc != 0:
        raise NkpError(rc, lib.nkp_last_error().decode())
    return rp, ci[:nloc], v[:nloc]


class NkpSolver:
    """Device-resident solver for one CSR matrix (setup = the reference's factor-only call)."""

    def __init__(self, rowptr, colind, val, blk_start=None, coupled_tracer_cnt=1, col_i=None, col_j=None, col_t=None, **options):
        self._lib = load_library()
        self._h = C.c_void_p()
        # 64-bit row pointers (CDF-5 matrix files, callers counting entries in int64) go through nkp_create64
        self._rowptr64 = np.ascontiguousarray(rowptr, np.int64) if np.asarray(rowptr).dtype == np.int64 else None
        rowptr = np.ascontiguousarray(rowptr, np.int32) if self._rowptr64 is None else self._rowptr64
        colind = np.ascontiguousarray(colind, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        self.n = int(rowptr.size - 1)
        self.nnz = int(colind.size)
        tuning = options.pop("tuning", None)
        opt = default_options(**options)
        self.options = opt
        if tuning is not None:                                 # an NkpTuning, or a dict of overrides on top of the defaults + environment
            self._tuning = tuning if isinstance(tuning, NkpTuning) else default_tuning(**tuning)
            opt.tuning = C.pointer(self._tuning)
        if col_i is not None and col_j is not None:
            col_i = np.ascontiguousarray(col_i, np.int32)
            col_j = np.ascontiguousarray(col_j, np.int32)
            if blk_start is None or col_i.size != len(blk_start) - 1 or col_j.size != col_i.size:
                raise ValueError("col_i / col_j need one entry per block of blk_start")
            opt.col_i, opt.col_j = _p(col_i, C.c_int32), _p(col_j, C.c_int32)
        if col_t is not None:                                  # rows not tracer-major (cell_major_order): tracer of every block
            col_t = np.ascontiguousarray(col_t, np.int32)
            if blk_start is None or col_t.size != len(blk_start) - 1:
                raise ValueError("col_t needs one entry per block of blk_start")
            opt.col_t = _p(col_t, C.c_int32)
        if blk_start is not None:
            blk_start = np.ascontiguousarray(blk_start, np.int32)
            bp, nb = _p(blk_start, C.c_int32), blk_start.size - 1
        else:
            bp, nb = None, 0
        if self._rowptr64 is not None:
            rc = self._lib.nkp_create64(C.byref(self._h), C.byref(opt), self.n, _p(self._rowptr64, C.c_int64),
                                        _p(colind, C.c_int32), _p(val, C.c_double), bp, nb, coupled_tracer_cnt)
        else:
            rc = self._lib.nkp_create(C.byref(self._h), C.byref(opt), self.n, self.nnz, _p(rowptr, C.c_int32),
                                      _p(colind, C.c_int32), _p(val, C.c_double), bp, nb, coupled_tracer_cnt)
        if rc != 0:
            self._h = C.c_void_p()
            raise NkpError(rc, self._lib.nkp_last_error().decode())

    @classmethod
    def _from_handle(cls, lib, handle, n, nnz, options):
        self = cls.__new__(cls)
        self._lib, self._h, self.n, self.nnz, self.options = lib, handle, n, nnz, options
        return self

    def _check(self, rc, allow=(0,)):
        if rc not in allow:
            raise NkpError(rc, self._lib.nkp_last_error().decode())
        return rc

    def solve(self, b, raise_on_fail=True):
        """Returns (x, info) with info = dict(status, iters, relres, berr); b is not modified."""
        x = np.array(b, np.float64, order="C", copy=True).reshape(-1)
        if x.size != self.n:
            raise ValueError(f"b has {x.size} entries, expected {self.n}")
        berr, relres, iters = C.c_double(), C.c_double(), C.c_int()
        rc = self._lib.nkp_solve(self._h, _p(x, C.c_double), 1, self.n, C.byref(berr), C.byref(iters), C.byref(relres))
        self._check(rc, (0,) if raise_on_fail else (0, 1, 2, 3))
        return x, dict(status=rc, iters=iters.value, relres=relres.value, berr=berr.value)

    def solve_device(self, d_b, d_x, use_guess=False, raise_on_fail=True):
        """d_b / d_x: integer device addresses (e.g. torch tensor .data_ptr()) of n float64."""
        berr, relres, iters = C.c_double(), C.c_double(), C.c_int()
        rc = self._lib.nkp_solve_device(self._h, C.c_void_p(d_b), C.c_void_p(d_x), int(use_guess), C.byref(berr),
                                        C.byref(iters), C.byref(relres))
        self._check(rc, (0,) if raise_on_fail else (0, 1, 2, 3))
        return dict(status=rc, iters=iters.value, relres=relres.value, berr=berr.value)

    def solve_batch_device(self, d_B, d_X, nrhs, ldb, raise_on_fail=True):
        """nkp_solve_batch_device: nrhs right-hand sides resident on the device (vector c at d_B + 8 * c * ldb), solutions to d_X.
        Returns one info dict per right-hand side."""
        berr, relres, iters = (C.c_double * nrhs)(), (C.c_double * nrhs)(), (C.c_int * nrhs)()
        rc = self._lib.nkp_solve_batch_device(self._h, nrhs, C.c_void_p(d_B), C.c_void_p(d_X), ldb, berr, iters, relres)
        self._check(rc, (0,) if raise_on_fail else (0, 1, 2, 3))
        return [dict(status=rc, iters=iters[c], relres=relres[c], berr=berr[c]) for c in range(nrhs)]

    def solve_many(self, B, raise_on_fail=True):
        """nkp_solve with nrhs = B.shape[0] host right-hand sides (rows of B); returns (X, infos)."""
        X = np.array(B, np.float64, order="C", copy=True)
        nrhs = X.shape[0]
        berr, relres, iters = (C.c_double * nrhs)(), (C.c_double * nrhs)(), (C.c_int * nrhs)()
        rc = self._lib.nkp_solve(self._h, _p(X, C.c_double), nrhs, X.shape[1], berr, iters, relres)
        self._check(rc, (0,) if raise_on_fail else (0, 1, 2, 3))
        return X, [dict(status=rc, iters=iters[c], relres=relres[c], berr=berr[c]) for c in range(nrhs)]

    def spmv(self, x):
        x = np.ascontiguousarray(x, np.float64)
        y = np.empty(self.n)
        self._check(self._lib.nkp_spmv(self._h, _p(x, C.c_double), _p(y, C.c_double)))
        return y

    def spmv_device(self, d_x, d_y):
        self._check(self._lib.nkp_spmv_device(self._h, C.c_void_p(d_x), C.c_void_p(d_y)))

    def precond_apply(self, r):
        r = np.ascontiguousarray(r, np.float64)
        z = np.empty(self.n)
        self._check(self._lib.nkp_precond_apply(self._h, _p(r, C.c_double), _p(z, C.c_double)))
        return z

    def multi_dot(self, V, w):
        V = np.ascontiguousarray(V, np.float64)
        w = np.ascontiguousarray(w, np.float64)
        k = V.shape[0]
        out = np.empty(k + 1)
        self._check(self._lib.nkp_multi_dot(self._h, _p(V, C.c_double), V.shape[1], k, _p(w, C.c_double), _p(out, C.c_double)))
        return out

    def time_kernel(self, which, reps=20, arg=0):
        ms = C.c_double()
        self._check(self._lib.nkp_time_kernel(self._h, which, arg, reps, C.byref(ms)))
        return ms.value

    def get_int(self, key):
        return int(self._lib.nkp_get_int(self._h, key.encode()))

    _ML_ARRAY_TYPES = {"valf": np.float32, "val": np.float64, "fac": np.float64, "coarse_inv": np.float64}

    def ml_level_array(self, level, what):
        """One array of the multilevel hierarchy as it sits on the device (nkp_ml_level_array); empty if the level has none."""
        cnt = int(self._lib.nkp_ml_level_array(self._h, level, what.encode(), None, 0))
        if cnt < 0:
            raise NkpError(cnt, last_error())
        out = np.empty(cnt, self._ML_ARRAY_TYPES.get(what, np.int32))
        if cnt:
            got = int(self._lib.nkp_ml_level_array(self._h, level, what.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes))
            if got != cnt:
                raise NkpError(got, last_error())
        return out

    def set_stream(self, stream_ptr):
        self._check(self._lib.nkp_set_stream(self._h, C.c_void_p(stream_ptr)))

    def clone(self):
        """A second handle on the same device-resident matrix and hierarchy with its own work vectors and stream
        (nkp_clone): solve another right-hand side concurrently from another thread.  Close it before this one."""
        h = C.c_void_p()
        self._check(self._lib.nkp_clone(self._h, C.byref(h)))
        c = object.__new__(NkpSolver)
        c._lib, c._h, c.n, c.nnz, c.options, c._parent = self._lib, h, self.n, self.nnz, self.options, self
        return c

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.nkp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def device_count():
    return int(load_library().nkp_device_count())


def last_error():
    """Text of the most recent failure reported by the library on this thread (nkp_last_error)."""
    return load_library().nkp_last_error().decode()


def column_coords(ind_i, ind_j, col_start, coupled_tracer_cnt=1):
    """(i, j) of every block of column_blocks(...): the index maps at each column's first row."""
    first = np.asarray(col_start[:-1], np.int64)
    ci = np.tile(np.asarray(ind_i)[first], coupled_tracer_cnt).astype(np.int32)
    cj = np.tile(np.asarray(ind_j)[first], coupled_tracer_cnt).astype(np.int32)
    return ci, cj


def column_blocks(col_start, tracer_state_len, coupled_tracer_cnt=1):
    """blk_start for tracer-major rows (reference src/matrix.c:778-784) from one tracer's col_start."""
    col_start = np.asarray(col_start, np.int64)
    parts = [col_start[:-1] + t * tracer_state_len for t in range(coupled_tracer_cnt)]
    return np.concatenate(parts + [np.array([coupled_tracer_cnt * tracer_state_len])]).astype(np.int32)
