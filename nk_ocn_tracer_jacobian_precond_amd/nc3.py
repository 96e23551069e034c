"""NetCDF classic (CDF-1 / CDF-2 / CDF-5) reader + writer in pure numpy.

Host-side tooling only (synthetic-problem generator, tests, bench input staging).  The
product's file I/O is the C codec in ``host/nc3_codec.c``; this module is its independent
Python twin so tests can cross-check the two (and scipy.io.netcdf_file as a third opinion).

Format facts follow the on-disk contract the reference's writers produce: the matrix file is
created NC_64BIT_OFFSET, i.e. ``CDF\\x02`` (reference src/grid.c:235), all data big-endian.
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE = 1, 2, 3, 4, 5, 6
NC_UBYTE, NC_USHORT, NC_UINT, NC_INT64, NC_UINT64 = 7, 8, 9, 10, 11
NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C

_DT = {
    NC_BYTE: ">i1", NC_CHAR: "S1", NC_SHORT: ">i2", NC_INT: ">i4", NC_FLOAT: ">f4",
    NC_DOUBLE: ">f8", NC_UBYTE: ">u1", NC_USHORT: ">u2", NC_UINT: ">u4", NC_INT64: ">i8",
    NC_UINT64: ">u8",
}
_NP2NC = {
    "int8": NC_BYTE, "int16": NC_SHORT, "int32": NC_INT, "float32": NC_FLOAT,
    "float64": NC_DOUBLE, "uint8": NC_UBYTE, "uint16": NC_USHORT, "uint32": NC_UINT,
    "int64": NC_INT64, "uint64": NC_UINT64,
}


def _pad4(n):
    return (n + 3) & ~3


class NcVar:
    def __init__(self, name, nc_type, dims, atts, begin=0, vsize=0):
        self.name, self.nc_type, self.dims, self.atts = name, nc_type, dims, atts
        self.begin, self.vsize = begin, vsize
        self.shape = ()
        self.is_record = False


class NcFile:
    """Parsed header of a classic NetCDF file; data access is whole-variable like the
    reference's nc_get_var_* / nc_put_var_* use (reference src/file_io.c:72-368)."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            head = f.read(4)
            if head[:3] != b"CDF" or head[3] not in (1, 2, 5):
                if head == b"\x89HDF":
                    raise ValueError(f"{path}: NetCDF-4/HDF5 files are not supported (classic CDF-1/2/5 only)")
                raise ValueError(f"{path}: not a NetCDF classic file")
            self.version = head[3]
            self._f = f
            self._nn = 8 if self.version == 5 else 4      # NON_NEG width
            self._off = 4 if self.version == 1 else 8      # begin-offset width
            self.numrecs = self._nonneg()
            self.dims = OrderedDict()
            tag, cnt = self._int4(), self._nonneg()
            if tag == NC_DIMENSION:
                for _ in range(cnt):
                    name = self._name()
                    self.dims[name] = self._nonneg()
            self.gatts = self._attlist()
            self.vars = OrderedDict()
            tag, cnt = self._int4(), self._nonneg()
            dimnames = list(self.dims)
            if tag == NC_VARIABLE:
                for _ in range(cnt):
                    name = self._name()
                    nd = self._nonneg()
                    dimids = [self._nonneg() for _ in range(nd)]
                    atts = self._attlist()
                    nc_type = self._int4()
                    vsize = self._nonneg()
                    begin = struct.unpack(">i" if self._off == 4 else ">q", f.read(self._off))[0]
                    v = NcVar(name, nc_type, [dimnames[d] for d in dimids], atts, begin, vsize)
                    v.is_record = nd > 0 and self.dims[v.dims[0]] == 0
                    v.shape = tuple(self.dims[d] for d in v.dims)
                    self.vars[name] = v
            self.header_len = f.tell()
            del self._f
        recvars = [v for v in self.vars.values() if v.is_record]
        self.recsize = sum(v.vsize for v in recvars)
        if len(recvars) == 1:   # single record variable: records are packed without padding
            v = recvars[0]
            self.recsize = int(np.prod(v.shape[1:], dtype=np.int64)) * np.dtype(_DT[v.nc_type]).itemsize

    # -- header primitives -------------------------------------------------------------
    def _int4(self):
        return struct.unpack(">i", self._f.read(4))[0]

    def _nonneg(self):
        return struct.unpack(">i" if self._nn == 4 else ">q", self._f.read(self._nn))[0]

    def _name(self):
        n = self._nonneg()
        s = self._f.read(_pad4(n))[:n]
        return s.decode("utf-8")

    def _attlist(self):
        tag, cnt = self._int4(), self._nonneg()
        atts = OrderedDict()
        if tag == NC_ATTRIBUTE:
            for _ in range(cnt):
                name = self._name()
                t = self._int4()
                n = self._nonneg()
                nbytes = n * np.dtype(_DT[t]).itemsize
                raw = self._f.read(_pad4(nbytes))[:nbytes]
                atts[name] = raw.decode("utf-8", "replace") if t == NC_CHAR else np.frombuffer(raw, _DT[t]).copy()
        return atts

    # -- data --------------------------------------------------------------------------
    def full_shape(self, name):
        v = self.vars[name]
        return ((self.numrecs,) + v.shape[1:]) if v.is_record else v.shape

    def get(self, name):
        """Whole variable, native-endian numpy array in the file's own type."""
        v = self.vars[name]
        dt = np.dtype(_DT[v.nc_type])
        shape = self.full_shape(name)
        cnt = int(np.prod(shape, dtype=np.int64))
        with open(self.path, "rb") as f:
            if not v.is_record:
                f.seek(v.begin)
                a = np.fromfile(f, dt, cnt)
            else:
                per = int(np.prod(v.shape[1:], dtype=np.int64))
                a = np.empty(cnt, dt)
                for r in range(self.numrecs):
                    f.seek(v.begin + r * self.recsize)
                    a[r * per:(r + 1) * per] = np.fromfile(f, dt, per)
        if a.size != cnt:
            raise IOError(f"{self.path}:{name}: short read")
        return a.astype(dt.newbyteorder("=")).reshape(shape)

    def put(self, name, data):
        """Overwrite a whole variable in place (no redefinition), converting to the file type."""
        v = self.vars[name]
        dt = np.dtype(_DT[v.nc_type])
        shape = self.full_shape(name)
        a = np.ascontiguousarray(np.asarray(data).reshape(shape), dtype=dt)
        with open(self.path, "r+b") as f:
            if not v.is_record:
                f.seek(v.begin)
                a.tofile(f)
            else:
                for r in range(self.numrecs):
                    f.seek(v.begin + r * self.recsize)
                    a[r].tofile(f)


def write(path, dims, variables, gatts=None, version=2):
    """Create a classic file.

    dims: OrderedDict name -> length.  variables: list of (name, dimnames, array, atts) with
    atts a dict of str | numpy scalars/arrays.  No record dimension (the matrix file the
    reference writes has none: src/grid.c:240-247, src/matrix.c:291, 3868-3872).
    """
    nn = 8 if version == 5 else 4
    offw = 4 if version == 1 else 8

    def nonneg(x):
        return struct.pack(">i" if nn == 4 else ">q", int(x))

    def name(s):
        b = s.encode("utf-8")
        return nonneg(len(b)) + b + b"\0" * (_pad4(len(b)) - len(b))

    def attlist(atts):
        if not atts:
            return struct.pack(">i", 0) + nonneg(0)
        out = struct.pack(">i", NC_ATTRIBUTE) + nonneg(len(atts))
        for k, val in atts.items():
            if isinstance(val, str):
                raw, t, n = val.encode("utf-8"), NC_CHAR, len(val.encode("utf-8"))
            else:
                arr = np.atleast_1d(np.asarray(val))
                t = _NP2NC[arr.dtype.name]
                raw, n = arr.astype(_DT[t]).tobytes(), arr.size
            out += name(k) + struct.pack(">i", t) + nonneg(n) + raw + b"\0" * (_pad4(len(raw)) - len(raw))
        return out

    dimnames = list(dims)
    vars_meta = []
    for vname, vdims, arr, atts in variables:
        arr = np.asarray(arr)
        t = _NP2NC[arr.dtype.name]
        shape = tuple(dims[d] for d in vdims)
        if int(np.prod(shape, dtype=np.int64)) != arr.size:
            raise ValueError(f"{vname}: shape mismatch {shape} vs {arr.shape}")
        nbytes = arr.size * np.dtype(_DT[t]).itemsize
        vars_meta.append((vname, vdims, arr, atts, t, _pad4(nbytes)))

    def header(begins):
        h = b"CDF" + bytes([version]) + nonneg(0)
        if dims:
            h += struct.pack(">i", NC_DIMENSION) + nonneg(len(dims))
            for d, n in dims.items():
                h += name(d) + nonneg(n)
        else:
            h += struct.pack(">i", 0) + nonneg(0)
        h += attlist(gatts)
        if vars_meta:
            h += struct.pack(">i", NC_VARIABLE) + nonneg(len(vars_meta))
            for (vname, vdims, arr, atts, t, vsize), b in zip(vars_meta, begins):
                h += name(vname) + nonneg(len(vdims))
                for d in vdims:
                    h += nonneg(dimnames.index(d))
                # CDF-1/2 store vsize in 32 bits; oversize variables write 2^32-1 per the spec
                h += attlist(atts) + struct.pack(">i", t)
                h += struct.pack(">I", min(vsize, 0xFFFFFFFF)) if nn == 4 else struct.pack(">q", vsize)
                h += struct.pack(">i" if offw == 4 else ">q", b)
        else:
            h += struct.pack(">i", 0) + nonneg(0)
        return h

    hlen = len(header([0] * len(vars_meta)))
    begins, pos = [], _pad4(hlen)
    for m in vars_meta:
        begins.append(pos)
        pos += m[5]
    h = header(begins)
    with open(path, "wb") as f:
        f.write(h)
        f.write(b"\0" * (_pad4(hlen) - hlen))
        for (vname, vdims, arr, atts, t, vsize), b in zip(vars_meta, begins):
            assert f.tell() == b
            a = np.ascontiguousarray(arr, dtype=_DT[t])
            a.tofile(f)
            pad = vsize - a.nbytes
            if pad:
                f.write(b"\0" * pad)
