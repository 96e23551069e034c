"""NetCDF classic codec: python twin vs scipy.io.netcdf_file vs the C codec behind the
reference-named file_io surface (pin p1 of SURVEY.md section 8c)."""
import ctypes as C
import os
import shutil
from collections import OrderedDict

import numpy as np
import pytest

from nk_ocn_tracer_jacobian_precond_amd import nc3, solver, synth


@pytest.fixture(scope="module")
def host():
    L = C.CDLL(solver.HOST_LIB_PATH)
    L.malloc_3d_double.restype = C.POINTER(C.POINTER(C.POINTER(C.c_double)))
    L.malloc_3d_double.argtypes = [C.c_int] * 3
    L.free_3d_double.argtypes = [C.POINTER(C.POINTER(C.POINTER(C.c_double)))]
    return L


def test_python_codec_matches_scipy(golden):
    from scipy.io import netcdf_file
    f = netcdf_file(golden.matrix_path, "r", mmap=False)
    g = nc3.NcFile(golden.matrix_path)
    assert f.version_byte == 2 == g.version          # NC_64BIT_OFFSET like reference src/grid.c:235
    assert dict(f.dimensions) == dict(g.dims)
    for name in ("rowptr", "colind", "nzval_row_wise", "KMT", "int3_to_tracer_state_ind", "TLAT", "z_t"):
        assert np.array_equal(np.asarray(f.variables[name][:]), g.get(name)), name
    assert int(f.variables["coupled_tracer_cnt"].getValue()) == golden.cnt
    assert g.vars["int3_to_tracer_state_ind"].atts["_FillValue"][0] == -1
    f.close()


def test_c_reader_matches_python(golden, host):
    assert host.get_sparse_matrix(golden.matrix_path.encode()) == 0
    n = C.c_int.in_dll(host, "flat_len").value
    nnz = C.c_int.in_dll(host, "nnz").value
    assert (n, nnz) == (golden.n, golden.colind.size)
    assert C.c_int.in_dll(host, "coupled_tracer_cnt").value == golden.cnt
    val = np.ctypeslib.as_array(C.POINTER(C.c_double).in_dll(host, "nzval_row_wise"), (nnz,))
    ci = np.ctypeslib.as_array(C.POINTER(C.c_int).in_dll(host, "colind"), (nnz,))
    rp = np.ctypeslib.as_array(C.POINTER(C.c_int).in_dll(host, "rowptr"), (n + 1,))
    assert np.array_equal(val, golden.val) and np.array_equal(ci, golden.colind) and np.array_equal(rp, golden.rowptr)
    host.free_sparse_matrix()
    assert host.get_ind_maps(golden.matrix_path.encode()) == 0
    assert (C.c_int.in_dll(host, "imt").value, C.c_int.in_dll(host, "jmt").value, C.c_int.in_dll(host, "km").value) == (golden.imt, golden.jmt, golden.km)
    assert C.c_int.in_dll(host, "tracer_state_len").value == golden.tsl
    host.nkp_column_blocks.restype = C.POINTER(C.c_int)
    nb = C.c_int()
    cs = host.nkp_column_blocks(C.byref(nb))
    assert np.array_equal(np.ctypeslib.as_array(cs, (nb.value + 1,)), golden.blk_start)
    host.free_ind_maps()


@pytest.mark.parametrize("version", [1, 2, 5])
@pytest.mark.parametrize("nc_type", ["float64", "float32"])
def test_put_get_roundtrip_in_place(tmp_path, host, version, nc_type):
    """put_var_3d_double overwrites the variable in place and get_var_3d_double converts any
    numeric external type to double (libnetcdf semantics the reference relies on)."""
    p = synth.generate(imt=6, jmt=5, km=4, seed=1)
    fields = synth.make_tracer_fields(p, ["A", "B"], seed=3)
    path = str(tmp_path / "t.nc")
    synth.write_tracer_file(p, path, fields, version=version, nc_type=nc_type)
    cube = host.malloc_3d_double(p.km, p.jmt, p.imt)
    assert host.get_var_3d_double(path.encode(), b"B", cube) == 0
    flat = np.ctypeslib.as_array(cube[0][0], (p.km * p.jmt * p.imt,))
    want = fields["B"].astype(nc_type).astype(np.float64).reshape(-1)
    assert np.array_equal(flat, want)
    flat[:] = np.arange(flat.size) * 0.5
    assert host.put_var_3d_double(path.encode(), b"B", cube) == 0
    g = nc3.NcFile(path)
    assert np.array_equal(g.get("B").reshape(-1), (np.arange(flat.size) * 0.5).astype(nc_type))
    assert np.array_equal(g.get("A"), fields["A"].astype(nc_type))          # neighbour variable untouched
    host.free_3d_double(cube)


def test_record_variable(tmp_path, host):
    """CESM tracer files carry a time record dimension; a 1-record variable reads as km*jmt*imt."""
    from scipy.io import netcdf_file
    path = str(tmp_path / "rec.nc")
    f = netcdf_file(path, "w", version=2)
    f.createDimension("time", None)
    f.createDimension("z_t", 3)
    f.createDimension("nlat", 4)
    f.createDimension("nlon", 5)
    v = f.createVariable("IAGE", "d", ("time", "z_t", "nlat", "nlon"))
    w = f.createVariable("OTHER", "f", ("time", "nlat", "nlon"))
    data = np.random.default_rng(0).standard_normal((1, 3, 4, 5))
    v[0] = data[0]
    w[0] = np.ones((4, 5), np.float32)
    f.close()
    g = nc3.NcFile(path)
    assert g.numrecs == 1 and np.array_equal(g.get("IAGE"), data)
    cube = host.malloc_3d_double(3, 4, 5)
    nel = C.c_size_t()
    assert host.nkp_var_nelems(path.encode(), b"IAGE", C.byref(nel)) == 0 and nel.value == 60
    assert host.get_var_3d_double(path.encode(), b"IAGE", cube) == 0
    assert np.array_equal(np.ctypeslib.as_array(cube[0][0], (60,)), data.reshape(-1))
    host.free_3d_double(cube)


def test_errors(tmp_path, host, capfd):
    missing = str(tmp_path / "nope.nc").encode()
    assert host.get_grid_dims(missing) != 0
    assert "ERROR returned from netCDF routine" in capfd.readouterr().err
    hdf = tmp_path / "h.nc"
    hdf.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    assert host.get_grid_dims(str(hdf).encode()) != 0
    assert "HDF5" in capfd.readouterr().err
    g = os.path.join(os.path.dirname(__file__), "golden", "tri_12x10x6_tracers.nc")
    dst = str(tmp_path / "t.nc")
    shutil.copy(g, dst)
    buf = (C.c_double * 720)()
    assert host.get_var_1d_double(dst.encode(), b"NOT_THERE", buf) != 0
    assert "Variable not found" in capfd.readouterr().err
    exists = C.c_int(-1)
    assert host.var_exists_in_file(dst.encode(), b"IAGE", C.byref(exists)) == 0 and exists.value == 1
    assert host.var_exists_in_file(dst.encode(), b"NOT_THERE", C.byref(exists)) == 0 and exists.value == 0
    fv = C.c_double()
    assert host.get_att_double(dst.encode(), b"IAGE", b"_FillValue", C.byref(fv)) == 0
    assert fv.value == synth.FILL_DOUBLE


def test_row_slices_of_the_matrix_file(golden, host, capfd):
    """What a rank of solve_ABdist reads when there are several (get_sparse_matrix_header + get_sparse_matrix_rows: the
    reference's rowptr / colind / nzval slices, src/solve_ABdist.c:141-225, read straight from the file as hyperslabs):
    any row range gives exactly the entries of those rows; ranges outside the variable are refused."""
    path = golden.matrix_path.encode()
    assert host.get_sparse_matrix_header(path) == 0
    n = C.c_int.in_dll(host, "flat_len").value
    assert n == golden.n and C.c_int.in_dll(host, "nnz").value == golden.colind.size
    assert not C.POINTER(C.c_int).in_dll(host, "colind")                 # header only: no entry arrays
    rp = np.ctypeslib.as_array(C.POINTER(C.c_int).in_dll(host, "rowptr"), (n + 1,)).copy()
    assert np.array_equal(rp, golden.rowptr)
    host.get_sparse_matrix_rows.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    for r0, r1 in ((0, n), (0, 1), (n // 3, 2 * n // 3), (n - 1, n), (5, 5)):
        cnt = int(rp[r1] - rp[r0])
        ci, v = np.full(cnt + 1, -7, np.int32), np.full(cnt + 1, np.nan)
        assert host.get_sparse_matrix_rows(path, r0, r1, ci.ctypes.data_as(C.POINTER(C.c_int)), v.ctypes.data_as(C.POINTER(C.c_double))) == 0
        assert np.array_equal(ci[:cnt], golden.colind[rp[r0]:rp[r1]]) and np.array_equal(v[:cnt], golden.val[rp[r0]:rp[r1]])
        assert ci[cnt] == -7                                             # nothing written past the range
    assert host.get_sparse_matrix_rows(path, 3, n + 1, None, None) == 1
    assert "outside the matrix" in capfd.readouterr().err
    host.free_sparse_matrix()
    # the codec's range check (libnetcdf's NC_EEDGE)
    host.get_vara_1d_int.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    buf = np.zeros(4, np.int32)
    assert host.get_vara_1d_int(path, b"rowptr", n, 2, buf.ctypes.data_as(C.POINTER(C.c_int))) != 0
    assert "exceeds dimension bound" in capfd.readouterr().err
    assert host.get_vara_1d_int(path, b"rowptr", n - 1, 2, buf.ctypes.data_as(C.POINTER(C.c_int))) == 0
    assert list(buf[:2]) == list(golden.rowptr[n - 1:n + 1])
