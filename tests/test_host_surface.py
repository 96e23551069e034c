"""Host logic with the reference's names: parsers, slab allocators, gather/scatter, partition."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ora
from nk_ocn_tracer_jacobian_precond_amd import solver


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(solver.HOST_LIB_PATH)
    L.malloc_3d_double.restype = C.POINTER(C.POINTER(C.POINTER(C.c_double)))
    L.malloc_3d_double.argtypes = [C.c_int] * 3
    L.malloc_3d_int.restype = C.POINTER(C.POINTER(C.POINTER(C.c_int)))
    L.malloc_3d_int.argtypes = [C.c_int] * 3
    L.malloc_2d_double.restype = C.POINTER(C.POINTER(C.c_double))
    L.malloc_2d_double.argtypes = [C.c_int] * 2
    return L


@pytest.mark.parametrize("text,ok,val", [("12", True, 12), ("0x10", True, 16), ("010", True, 8), ("-7", True, -7),
                                          ("", False, 0), ("12a", False, 0), ("99999999999999999999", False, 0)])
def test_parse_to_long(host, text, ok, val, capfd):
    """strtol base 0, reject empty / trailing characters / ERANGE (reference src/misc.c:11-38)."""
    out = C.c_long()
    rc = host.parse_to_long(text.encode(), C.byref(out))
    assert (rc == 0) == ok
    if ok:
        assert out.value == val
    else:
        assert "parse_to_long" in capfd.readouterr().err


def test_parse_to_int_and_double(host, capfd):
    i = C.c_int()
    assert host.parse_to_int(b"2147483647", C.byref(i)) == 0 and i.value == 2147483647
    assert host.parse_to_int(b"2147483648", C.byref(i)) == 1          # out of int range (src/misc.c:57-60)
    assert "out of int range" in capfd.readouterr().err
    d = C.c_double()
    assert host.parse_to_double(b"1.5e-3", C.byref(d)) == 0 and d.value == 1.5e-3
    assert host.parse_to_double(b"1.5x", C.byref(d)) == 1
    assert host.parse_to_double(None, C.byref(d)) == 1


def test_slab_allocators_are_contiguous(host):
    """field[0][0] must address one contiguous km*jmt*imt slab (reference src/memory.c:60,67,135,142)."""
    km, jmt, imt = 3, 4, 5
    cube = host.malloc_3d_double(km, jmt, imt)
    base = C.addressof(cube[0][0].contents)
    for k in range(km):
        for j in range(jmt):
            assert C.addressof(cube[k][j].contents) == base + 8 * ((k * jmt + j) * imt)
    host.free_3d_double(cube)
    icube = host.malloc_3d_int(km, jmt, imt)
    assert C.addressof(icube[2][3].contents) == C.addressof(icube[0][0].contents) + 4 * ((2 * jmt + 3) * imt)
    host.free_3d_int(icube)
    m = host.malloc_2d_double(jmt, imt)
    assert C.addressof(m[3].contents) == C.addressof(m[0].contents) + 8 * 3 * imt
    host.free_2d_double(m)


def test_flatten_unflatten_preserve_land(golden, host):
    """B[t*tsl+s] = field[k][j][i]; the scatter touches ocean cells only (pin p2;
    reference src/solve_ABglobal.c:184-191, 236-248)."""
    assert host.get_sparse_matrix(golden.matrix_path.encode()) == 0
    assert host.get_ind_maps(golden.matrix_path.encode()) == 0
    cube = host.malloc_3d_double(golden.km, golden.jmt, golden.imt)
    flat = np.ctypeslib.as_array(cube[0][0], (golden.km * golden.jmt * golden.imt,))
    B = np.zeros(golden.n)
    host.nkp_flatten_tracer.argtypes = [C.c_int, C.c_void_p, np.ctypeslib.ndpointer(np.float64)]
    host.nkp_unflatten_tracer.argtypes = [C.c_int, np.ctypeslib.ndpointer(np.float64), C.c_void_p]
    for t, v in enumerate(golden.varnames[:golden.cnt]):
        flat[:] = golden.fields[v].reshape(-1)
        host.nkp_flatten_tracer(t, cube, B)
    assert np.array_equal(B, golden.rhs(golden.varnames[0]))
    # oracle restatement agrees
    want0 = ora.flatten(golden.ind_i, golden.ind_j, golden.ind_k, golden.imt, golden.jmt, golden.fields[golden.varnames[0]])
    assert np.array_equal(B[:golden.tsl], want0)
    X = np.arange(golden.n, dtype=np.float64) + 0.25
    f0 = golden.fields[golden.varnames[0]]
    flat[:] = f0.reshape(-1)
    host.nkp_unflatten_tracer(0, X, cube)
    after = flat.reshape(f0.shape).copy()
    ocean = np.zeros(f0.shape, bool)
    ocean[golden.ind_k, golden.ind_j, golden.ind_i] = True
    assert np.array_equal(after[~ocean], f0[~ocean])                       # land: bit-for-bit untouched
    assert np.array_equal(after[golden.ind_k, golden.ind_j, golden.ind_i], X[:golden.tsl])
    assert np.array_equal(after, ora.unflatten(golden.ind_i, golden.ind_j, golden.ind_k, golden.imt, golden.jmt, X[:golden.tsl], f0))
    host.free_3d_double(cube)
    host.free_ind_maps()
    host.free_sparse_matrix()


@pytest.mark.parametrize("n,P", [(10, 1), (10, 2), (10, 3), (17, 8), (324, 7), (8, 8)])
def test_rowblock_partition_rule(host, n, P):
    """m_loc = n / P, last rank takes the remainder (reference src/solve_ABdist.c:141-144)."""
    covered = 0
    for rank in range(P):
        f, m = C.c_int(), C.c_int()
        host.nkp_rowblock_partition(n, P, rank, C.byref(f), C.byref(m))
        assert (f.value, m.value) == ora.rowblock_partition(n, P, rank)
        assert f.value == covered
        assert m.value == (n // P if rank < P - 1 else n - (P - 1) * (n // P))
        covered += m.value
    assert covered == n
