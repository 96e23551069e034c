"""Test helper: writes HDF5 files laid out the way the netCDF-4 library lays out a file (root-group datasets for variables
and dimensions, dimension placeholders with the NAME attribute netCDF-4 gives them, _FillValue attributes of the variable's
type, optionally chunked + deflated), through ctypes on the libhdf5 of this image.  No netCDF library exists here to
produce a real one; the reader under test (host/nc4_hdf5.c) only relies on the layout rules listed in its header."""
import ctypes as C
import os

import numpy as np

_lib = None
hid_t = C.c_int64
hsize_t = C.c_ulonglong


def lib():
    global _lib
    if _lib is None:
        for name in (os.environ.get("NKP_HDF5_LIB"), "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so"):
            if not name:
                continue
            try:
                _lib = C.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            return None
        L = _lib
        L.H5open()
        for fn, res, args in (("H5Fcreate", hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), ("H5Fclose", C.c_int, [hid_t]),
                              ("H5Screate_simple", hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), ("H5Screate", hid_t, [C.c_int]),
                              ("H5Sclose", C.c_int, [hid_t]),
                              ("H5Dcreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                              ("H5Dwrite", C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), ("H5Dclose", C.c_int, [hid_t]),
                              ("H5Pcreate", hid_t, [hid_t]), ("H5Pset_chunk", C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
                              ("H5Pset_deflate", C.c_int, [hid_t, C.c_uint]), ("H5Pclose", C.c_int, [hid_t]),
                              ("H5Acreate2", hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), ("H5Awrite", C.c_int, [hid_t, hid_t, C.c_void_p]),
                              ("H5Aclose", C.c_int, [hid_t]), ("H5Tcopy", hid_t, [hid_t]), ("H5Tset_size", C.c_int, [hid_t, C.c_size_t]),
                              ("H5Tclose", C.c_int, [hid_t])):
            f = getattr(L, fn)
            f.restype, f.argtypes = res, args
    return _lib


def _g(name):
    return hid_t.in_dll(lib(), name).value


def _types():
    return {np.dtype("float64"): _g("H5T_IEEE_F64LE_g"), np.dtype("float32"): _g("H5T_IEEE_F32LE_g"), np.dtype("int32"): _g("H5T_STD_I32LE_g"),
            np.dtype("int16"): _g("H5T_STD_I16LE_g")}


def write(path, dims, variables, deflate=()):
    """dims: {name: length}; variables: [(name, [dim names], array, fill or None)].  Variables named in `deflate` are stored
    chunked and compressed."""
    L = lib()
    T = _types()
    fid = L.H5Fcreate(path.encode(), 2, 0, 0)              # H5F_ACC_TRUNC
    assert fid >= 0

    def dataset(name, arr, fill=None, compress=False, note=None):
        arr = np.ascontiguousarray(arr)
        shape = (hsize_t * max(arr.ndim, 1))(*arr.shape) if arr.ndim else None
        sp = L.H5Screate_simple(arr.ndim, shape, None) if arr.ndim else L.H5Screate(0)     # H5S_SCALAR
        dcpl = 0
        if compress and arr.ndim:
            dcpl = L.H5Pcreate(_g("H5P_CLS_DATASET_CREATE_ID_g"))
            chunk = (hsize_t * arr.ndim)(*[max(1, (s + 1) // 2) for s in arr.shape])
            assert L.H5Pset_chunk(dcpl, arr.ndim, chunk) >= 0 and L.H5Pset_deflate(dcpl, 4) >= 0
        d = L.H5Dcreate2(fid, name.encode(), T[arr.dtype], sp, 0, dcpl, 0)
        assert d >= 0, name
        assert L.H5Dwrite(d, T[arr.dtype], 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0
        if fill is not None:
            one = L.H5Screate(0)
            a = L.H5Acreate2(d, b"_FillValue", T[arr.dtype], one, 0, 0)
            v = np.array([fill], arr.dtype)
            assert L.H5Awrite(a, T[arr.dtype], v.ctypes.data_as(C.c_void_p)) >= 0
            L.H5Aclose(a)
            L.H5Sclose(one)
        if note is not None:
            st = L.H5Tcopy(_g("H5T_C_S1_g"))
            L.H5Tset_size(st, len(note) + 1)
            one = L.H5Screate(0)
            a = L.H5Acreate2(d, b"NAME", st, one, 0, 0)
            L.H5Awrite(a, st, C.create_string_buffer(note.encode()))
            L.H5Aclose(a)
            L.H5Sclose(one)
            L.H5Tclose(st)
        if dcpl:
            L.H5Pclose(dcpl)
        L.H5Sclose(sp)
        L.H5Dclose(d)

    names = {v[0] for v in variables}
    for dname, dlen in dims.items():
        if dname not in names:                             # a dimension without a coordinate variable: netCDF-4's placeholder dataset
            dataset(dname, np.zeros(dlen, np.float32), note=f"This is a netCDF dimension but not a netCDF variable.{dlen:>10d}")
    for name, vdims, arr, fill in variables:
        arr = np.asarray(arr)
        assert tuple(arr.shape) == tuple(dims[d] for d in vdims), name
        dataset(name, arr, fill, compress=name in deflate)
    L.H5Fclose(fid)
