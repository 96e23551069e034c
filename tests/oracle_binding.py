"""ctypes binding of oracle/libnkp_oracle.so -- TEST INFRASTRUCTURE (see oracle/nkp_oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_ROOT, "oracle", "libnkp_oracle.so")
_lib = None

i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            import subprocess
            subprocess.run(["make", "-C", os.path.dirname(LIB_PATH)], check=True)
        L = C.CDLL(LIB_PATH)
        L.ora_spmv.argtypes = [C.c_int64, i32p, i32p, f64p, f64p, f64p]
        L.ora_berr.argtypes = [C.c_int64, i32p, i32p, f64p, f64p, f64p]
        L.ora_berr.restype = C.c_double
        L.ora_flatten.argtypes = [C.c_int64, i32p, i32p, i32p, C.c_int, C.c_int, f64p, f64p]
        L.ora_unflatten.argtypes = [C.c_int64, i32p, i32p, i32p, C.c_int, C.c_int, f64p, f64p]
        L.ora_rowblock_partition.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.ora_localize_rowptr.argtypes = [C.c_int64, i32p]
        L.ora_colblock_measure.argtypes = [i32p, i32p, f64p, C.c_int64, i32p, i32p]
        L.ora_colblock_factor.argtypes = [C.c_int64, i32p, i32p, f64p, C.c_int64, i32p, C.c_int, f64p, C.POINTER(C.c_int)]
        L.ora_colblock_factor.restype = C.c_int64
        L.ora_colblock_apply.argtypes = [C.c_int64, C.c_int64, i32p, C.c_int, f64p, f64p, f64p]
        L.ora_multi_dot.argtypes = [C.c_int64, f64p, C.c_int64, C.c_int, f64p, f64p]
        L.ora_fgmres.argtypes = [C.c_int64, i32p, i32p, f64p, C.c_int64, i32p, C.c_int, C.c_int, C.c_int, C.c_double,
                                 f64p, f64p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ora_direct_solve.argtypes = [C.c_int64, i32p, i32p, f64p, f64p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.ora_num_threads.restype = C.c_int
        L.ora_ml_setup.argtypes = [C.c_int64, i32p, i32p, f64p, C.c_int64, i32p, i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ora_ml_setup.restype = C.c_void_p
        L.ora_ml_free.argtypes = [C.c_void_p]
        L.ora_ml_levels.argtypes = [C.c_void_p]
        L.ora_ml_level_rows.argtypes = [C.c_void_p, C.c_int]
        L.ora_ml_level_rows.restype = C.c_int64
        L.ora_ml_level_nnz.argtypes = [C.c_void_p, C.c_int]
        L.ora_ml_level_nnz.restype = C.c_int64
        L.ora_ml_level_maps.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.ora_ml_apply.argtypes = [C.c_void_p, f64p, f64p]
        L.ora_ml_fgmres.argtypes = [C.c_void_p, C.c_int64, i32p, i32p, f64p, C.c_int, C.c_int, C.c_double, C.c_int, f64p, f64p,
                                    C.POINTER(C.c_int), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _csr(rowptr, colind, val):
    return (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(colind, np.int32),
            np.ascontiguousarray(val, np.float64))


def spmv(rowptr, colind, val, x):
    rp, ci, v = _csr(rowptr, colind, val)
    y = np.empty(rp.size - 1)
    lib().ora_spmv(rp.size - 1, rp, ci, v, np.ascontiguousarray(x, np.float64), y)
    return y


def berr(rowptr, colind, val, x, b):
    rp, ci, v = _csr(rowptr, colind, val)
    return lib().ora_berr(rp.size - 1, rp, ci, v, np.ascontiguousarray(x, np.float64), np.ascontiguousarray(b, np.float64))


def direct_solve(rowptr, colind, val, b):
    rp, ci, v = _csr(rowptr, colind, val)
    x = np.array(b, np.float64, copy=True)
    be, steps = C.c_double(), C.c_int()
    info = lib().ora_direct_solve(rp.size - 1, rp, ci, v, x, C.byref(be), C.byref(steps))
    if info:
        raise RuntimeError(f"ora_direct_solve info={info}")
    return x, be.value


def colblock_measure(rowptr, colind, val, blk_start):
    rp, ci, v = _csr(rowptr, colind, val)
    bs = np.ascontiguousarray(blk_start, np.int32)
    out = np.zeros(3, np.int32)
    lib().ora_colblock_measure(rp, ci, v, bs.size - 1, bs, out)
    return tuple(int(t) for t in out)


def colblock_factor(rowptr, colind, val, blk_start, P):
    rp, ci, v = _csr(rowptr, colind, val)
    bs = np.ascontiguousarray(blk_start, np.int32)
    n = rp.size - 1
    fac = np.zeros((2 * P + 1) * n)
    dropped = C.c_int()
    bad = lib().ora_colblock_factor(n, rp, ci, v, bs.size - 1, bs, P, fac, C.byref(dropped))
    if bad:
        raise RuntimeError(f"zero pivot at row {bad - 1}")
    return fac, dropped.value


def colblock_apply(n, blk_start, P, fac, r):
    bs = np.ascontiguousarray(blk_start, np.int32)
    z = np.empty(n)
    lib().ora_colblock_apply(n, bs.size - 1, bs, P, fac, np.ascontiguousarray(r, np.float64), z)
    return z


def multi_dot(V, w):
    V = np.ascontiguousarray(V, np.float64)
    out = np.empty(V.shape[0] + 1)
    lib().ora_multi_dot(V.shape[1], V, V.shape[1], V.shape[0], np.ascontiguousarray(w, np.float64), out)
    return out


def fgmres(rowptr, colind, val, blk_start, b, precond=1, restart=100, max_iters=20000, rtol=1e-10):
    rp, ci, v = _csr(rowptr, colind, val)
    n = rp.size - 1
    bs = np.ascontiguousarray(blk_start if blk_start is not None else np.arange(n + 1), np.int32)
    x = np.zeros(n)
    it, rr = C.c_int(), C.c_double()
    # small systems: a handful of threads (128 OpenMP threads spinning on n = 300 take minutes)
    lib().ora_set_num_threads(int(max(1, min(os.cpu_count() or 1, n // 20000))))
    rc = lib().ora_fgmres(n, rp, ci, v, bs.size - 1, bs, precond, restart, max_iters, rtol,
                          np.ascontiguousarray(b, np.float64), x, C.byref(it), C.byref(rr))
    return x, dict(status=rc, iters=it.value, relres=rr.value)


def flatten(p_ind_i, p_ind_j, p_ind_k, imt, jmt, field):
    out = np.empty(p_ind_i.size)
    lib().ora_flatten(p_ind_i.size, p_ind_i, p_ind_j, p_ind_k, imt, jmt, np.ascontiguousarray(field, np.float64).reshape(-1), out)
    return out


def unflatten(p_ind_i, p_ind_j, p_ind_k, imt, jmt, B, field):
    f = np.ascontiguousarray(field, np.float64).reshape(-1).copy()
    lib().ora_unflatten(p_ind_i.size, p_ind_i, p_ind_j, p_ind_k, imt, jmt, np.ascontiguousarray(B, np.float64), f)
    return f.reshape(np.shape(field))


def rowblock_partition(n, nprocs, rank):
    a, b = C.c_int64(), C.c_int64()
    lib().ora_rowblock_partition(n, nprocs, rank, C.byref(a), C.byref(b))
    return a.value, b.value


def set_num_threads(t):
    lib().ora_set_num_threads(int(t))


def num_threads():
    return lib().ora_num_threads()


class MlOracle:
    """The multilevel water-column preconditioner + FGMRES of oracle/ml_oracle.c (C / OpenMP, all host cores)."""

    def __init__(self, rowptr, colind, val, blk_start, col_i, col_j, coupled_tracer_cnt=1, nu=3, coarsest_rows=3000, max_levels=0):
        self.rp, self.ci, self.v = _csr(rowptr, colind, val)
        self.n = self.rp.size - 1
        blk = np.ascontiguousarray(blk_start, np.int32)
        self._h = lib().ora_ml_setup(self.n, self.rp, self.ci, self.v, blk.size - 1, blk, np.ascontiguousarray(col_i, np.int32),
                                     np.ascontiguousarray(col_j, np.int32), coupled_tracer_cnt, nu, coarsest_rows, max_levels)
        if not self._h:
            raise RuntimeError("ora_ml_setup failed")

    def levels(self):
        L = lib()
        return [(int(L.ora_ml_level_rows(self._h, l)), int(L.ora_ml_level_nnz(self._h, l))) for l in range(L.ora_ml_levels(self._h))]

    def maps(self, l):
        rows = self.levels()[l][0]
        cmap, col_of = np.full(rows, -1, np.int32), np.empty(rows, np.int32)
        lib().ora_ml_level_maps(self._h, l, cmap, col_of)
        return cmap, col_of

    def apply(self, r):
        z = np.empty(self.n)
        lib().ora_ml_apply(self._h, np.ascontiguousarray(r, np.float64), z)
        return z

    def fgmres(self, b, restart=200, max_iters=20000, rtol=1e-10, reorth=0):
        x = np.empty(self.n)
        its, rr = C.c_int(), C.c_double()
        status = lib().ora_ml_fgmres(self._h, self.n, self.rp, self.ci, self.v, restart, max_iters, rtol, reorth,
                                     np.ascontiguousarray(b, np.float64), x, C.byref(its), C.byref(rr))
        return x, dict(status=status, iters=its.value, relres=rr.value)

    def close(self):
        if self._h:
            lib().ora_ml_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
