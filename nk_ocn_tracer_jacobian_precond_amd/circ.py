"""Synthetic circulation and tracer-source files for gen_A (host tooling: tests, examples).

gen_A reads a POP/CESM ocean history file (reference src/grid.c:90-213
and the load_* / add_* routines of src/matrix.c).  No such file exists offline, so this module
writes one with the same variable names, dimensions, staggering, units (POP cgs) and
`_FillValue` conventions, filled with smooth pseudo-random but physically scaled fields:

  grid     : z_t, dz, TLONG, TLAT, KMT, TAREA, DXU, DYU, HUS, HTE, HUW, HTN
  flow     : UVEL, VVEL (B-grid corner velocities, cm/s), WVEL (top-face, cm/s),
             UISOP, VISOP, WISOP (bolus), UTE_/VTN_/WTK_ POS/NEG (upwind3 face transports)
  mixing   : HDIF_EXPLICIT_3D_IRF_{1..4}_{1..3}_{1..3} (impulse responses of an isopycnal
             diffusion operator, 1/s), KAPPA_ISOP, HOR_DIFF, VDC_S, VDC_GM (cm^2/s),
             vmix_matrix_%03d_CUR (whole-column implicit mixing operator, 1/s)
  regions  : DYN_REGMASK (separate file)

and a tracer-source file with decay rates, linearised sources d_J_X_d_X[_k_NN], d_J_A_d_B,
piston velocities and surface-flux derivatives.  Nothing here runs inside the solver.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import nc3
from .synth import FILL_DOUBLE, _shift, _smooth2d, make_bathymetry, pop_like_dz

IRF_COLOURS = (4, 3, 3)


def _smooth3(rng, km, jmt, imt, passes=2):
    f = np.stack([_smooth2d(rng.standard_normal((jmt, imt)), passes) for _ in range(km)])
    f[1:-1] = 0.25 * (f[:-2] + f[2:]) + 0.5 * f[1:-1]
    return f / max(np.abs(f).max(), 1e-30)


def make_circulation(imt=12, jmt=10, km=6, seed=0, irf="diffusion", with_vmix_matrix=False, min_cos=0.3):
    """Return (fields, fills): name -> ndarray, and name -> fill value for masked variables."""
    rng = np.random.default_rng(seed + 77)
    dz = pop_like_dz(km)
    z_t = np.cumsum(dz) - 0.5 * dz
    KMT, lon, lat = make_bathymetry(imt, jmt, km, dz, seed)
    # a few negative KMT entries: the loader must treat them as land (src/grid.c:141-145)
    land = np.argwhere(KMT == 0)
    for (j, i) in land[:: max(1, len(land) // 3)][:3]:
        KMT[j, i] = -1
    KMTc = np.maximum(KMT, 0)
    R = 6.37122e8
    dlam, dphi = np.deg2rad(360.0 / imt), np.deg2rad(168.0 / jmt)
    cos_t = np.maximum(np.cos(np.deg2rad(lat)), min_cos)
    cos_u = np.maximum(np.cos(np.deg2rad(lat + 0.5 * 168.0 / jmt)), min_cos)
    wob = 1.0 + 0.05 * _smooth2d(rng.standard_normal((jmt, imt)), 1)          # mildly non-uniform metric
    DXT = (R * cos_t * dlam)[:, None] * wob
    DYT = np.full((jmt, imt), R * dphi) * wob[::-1]
    F = OrderedDict()
    F["z_t"], F["dz"] = z_t, dz
    F["TLONG"] = np.broadcast_to(lon[None, :], (jmt, imt)).copy()
    F["TLAT"] = np.broadcast_to(lat[:, None], (jmt, imt)).copy()
    F["KMT"] = KMT.astype(np.int32)
    F["TAREA"] = DXT * DYT
    F["DXU"] = (R * cos_u * dlam)[:, None] * wob
    F["DYU"] = DYT * 1.01
    F["HTN"] = F["DXU"] * 0.99           # north-face length of the T cell
    F["HTE"] = DYT * 1.02                # east-face length
    F["HUS"] = DXT * 1.01                # centre distance across the east face
    F["HUW"] = DYT * 0.98                # centre distance across the north face

    kk = np.arange(km)[:, None, None]
    M = kk < KMTc[None]
    KMTn = np.vstack([KMTc[1:], KMTc[-1:]])
    KMU = np.minimum(np.minimum(KMTc, np.roll(KMTc, -1, 1)), np.minimum(KMTn, np.roll(KMTn, -1, 1)))
    KMU[-1] = 0
    MU = kk < KMU[None]
    vert = (np.exp(-z_t / 80000.0) + 0.1)[:, None, None]
    fills = {}

    def masked(name, field, mask):
        F[name] = np.where(mask, field, FILL_DOUBLE)
        fills[name] = FILL_DOUBLE

    masked("UVEL", 5.0 * _smooth3(rng, km, jmt, imt) * vert, MU)
    masked("VVEL", 4.0 * _smooth3(rng, km, jmt, imt) * vert, MU)
    masked("WVEL", 2.0e-4 * _smooth3(rng, km, jmt, imt), M)
    masked("UISOP", 0.3 * _smooth3(rng, km, jmt, imt) * vert, M & np.roll(M, -1, 2))
    masked("VISOP", 0.3 * _smooth3(rng, km, jmt, imt) * vert, M & _shift(M, 0, 1, 0, False))
    masked("WISOP", 2.0e-5 * _smooth3(rng, km, jmt, imt), M)
    for nm in ("DXU", "DYU", "HTN", "HTE", "HUS", "HUW"):
        fills[nm] = FILL_DOUBLE

    # upwind3 inputs: face transports per unit depth split by sign (east / north faces), top-face w
    east_open = M & np.roll(M, -1, 2)
    north_open = M & _shift(M, 0, 1, 0, False)
    ute = 5.0 * _smooth3(rng, km, jmt, imt) * vert * F["HTE"][None] * east_open
    vtn = 4.0 * _smooth3(rng, km, jmt, imt) * vert * F["HTN"][None] * north_open
    wtk = 2.0e-4 * _smooth3(rng, km, jmt, imt) * (M & _shift(M, 0, 0, -1, False))
    for nm, f, msk in (("UTE", ute, east_open), ("VTN", vtn, north_open), ("WTK", wtk, M)):
        masked(nm + "_POS", np.maximum(f, 0.0), msk)
        masked(nm + "_NEG", np.minimum(f, 0.0), msk)

    # mixing coefficients
    mld = (3000.0 + 27000.0 * np.abs(np.sin(np.deg2rad(lat))) ** 3)[None, :, None]
    zbot = np.cumsum(dz)[:, None, None]
    masked("VDC_S", 0.1 + 1000.0 * (zbot < mld) + 0.05 * np.abs(_smooth3(rng, km, jmt, imt)), M)
    masked("VDC_GM", 5.0 * np.abs(_smooth3(rng, km, jmt, imt)), M)
    masked("KAPPA_ISOP", 6.0e6 * (0.5 + np.abs(_smooth3(rng, km, jmt, imt))), M)
    masked("HOR_DIFF", 3.0e6 * vert * np.abs(_smooth3(rng, km, jmt, imt)), M)

    # impulse-response fields of a (rotated) diffusion operator on the 15-point stencil
    ni, nj, nk = IRF_COLOURS
    irfs = np.zeros((ni, nj, nk, km, jmt, imt))
    if irf == "random":
        irfs = 1.0e-8 * rng.standard_normal(irfs.shape)
    else:
        kap = 6.0e6 * (0.5 + np.abs(_smooth3(rng, km, jmt, imt)))
        sx = 1.5e-3 * _smooth3(rng, km, jmt, imt) * np.clip(z_t / 3.0e4, 0, 1)[:, None, None]
        sy = 1.5e-3 * _smooth3(rng, km, jmt, imt) * np.clip(z_t / 3.0e4, 0, 1)[:, None, None]
        A = F["TAREA"][None]
        ce = kap * F["HTE"][None] / F["HUS"][None] / A * east_open
        cw = np.roll(kap * F["HTE"][None] / F["HUS"][None], 1, 2) / A * (M & np.roll(M, 1, 2))
        cn = kap * F["HTN"][None] / F["HUW"][None] / A * north_open
        cs = _shift(kap * F["HTN"][None] / F["HUW"][None], 0, -1, 0) / A * (M & _shift(M, 0, -1, 0, False))
        E = {(0, 0, 0): -(ce + cw + cn + cs), (1, 0, 0): ce, (-1, 0, 0): cw, (0, 1, 0): cn, (0, -1, 0): cs,
             (0, 0, -1): np.zeros_like(ce), (0, 0, 1): np.zeros_like(ce)}
        dzk = dz[:, None, None]
        for (d, cf, s) in (((1, 0), ce, sx), ((-1, 0), cw, -np.roll(sx, 1, 2)), ((0, 1), cn, sy), ((0, -1), cs, -_shift(sy, 0, -1, 0))):
            ok = M & _shift(M, 0, 0, -1, False) & _shift(M, 0, 0, 1, False) & _shift(M, d[0], d[1], -1, False) & _shift(M, d[0], d[1], 1, False)
            dh = (F["HUS"] if d[1] == 0 else F["HUW"])[None]
            x = 0.25 * cf * s * dh / dzk * ok
            E[(d[0], d[1], -1)] = x
            E[(d[0], d[1], 1)] = -x
            E[(0, 0, -1)] = E[(0, 0, -1)] + x
            E[(0, 0, 1)] = E[(0, 0, 1)] - x
        K, J, I = np.meshgrid(np.arange(km), np.arange(jmt), np.arange(imt), indexing="ij")
        for (di, dj, dk), val in E.items():
            ii, jj, k2 = (I + di) % imt, J + dj, K + dk
            ok = M & (jj >= 0) & (jj < jmt) & (k2 >= 0) & (k2 < km)
            np.add.at(irfs, ((ii % ni)[ok], (jj % nj)[ok], (k2 % nk)[ok], K[ok], J[ok], I[ok]), val[ok])
    for a in range(ni):
        for b in range(nj):
            for c in range(nk):
                F[f"HDIF_EXPLICIT_3D_IRF_{a + 1}_{b + 1}_{c + 1}"] = irfs[a, b, c]

    if with_vmix_matrix:
        # column operator: tridiagonal diffusion + a weak non-local (KPP-like) part from the top layers
        vdc = 0.1 + 50.0 * (zbot < mld)
        dzt = np.empty(km)
        dzt[0], dzt[1:] = dz[0], 0.5 * (dz[:-1] + dz[1:])
        ct = vdc / dzt[:, None, None] / dz[:, None, None] * (M & _shift(M, 0, 0, -1, False))
        cb = _shift(vdc / dzt[:, None, None], 0, 0, 1) / dz[:, None, None] * (M & _shift(M, 0, 0, 1, False))
        for kp in range(km):
            G = np.zeros((km, jmt, imt))
            G[kp] -= (ct + cb)[kp]
            if kp + 1 < km:
                G[kp + 1] += ct[kp + 1]
            if kp - 1 >= 0:
                G[kp - 1] += cb[kp - 1]
            if kp < 2:
                G += 1.0e-10 * np.abs(_smooth3(rng, km, jmt, imt)) * M * (kk > kp + 1)
            F[f"vmix_matrix_{kp + 1:03d}_CUR"] = G * (kp < KMTc[None])
    return F, fills


def write_circ_file(path, F, fills, nc_type="float64", version=2):
    km, (jmt, imt) = len(F["dz"]), F["KMT"].shape
    dims = OrderedDict([("nlon", imt), ("nlat", jmt), ("z_t", km)])
    variables = []
    for nm, a in F.items():
        a = np.asarray(a)
        vd = {1: ["z_t"], 2: ["nlat", "nlon"], 3: ["z_t", "nlat", "nlon"]}[a.ndim]
        if a.dtype.kind == "i":
            variables.append((nm, vd, a.astype(np.int32), None))
            continue
        t = "float64" if a.ndim == 1 else nc_type
        atts = OrderedDict([("_FillValue", np.asarray(fills[nm]).astype(t)[()])]) if nm in fills else None
        variables.append((nm, vd, a.astype(t), atts))
    nc3.write(path, dims, variables, version=version)


def write_region_file(path, KMT, seed=0):
    """DYN_REGMASK: negative = ignore (marginal seas); returns the mask."""
    rng = np.random.default_rng(seed + 5)
    jmt, imt = KMT.shape
    reg = np.ones((jmt, imt), np.int32)
    reg[KMT <= 0] = 0
    j0, i0 = rng.integers(1, jmt - 1), rng.integers(0, imt)
    reg[max(1, j0 - 1): j0 + 1, max(0, i0 - 2): i0 + 1] = -7
    nc3.write(path, OrderedDict([("nlon", imt), ("nlat", jmt)]), [("DYN_REGMASK", ["nlat", "nlon"], reg, None)])
    return reg


def make_tracer_sources(F, seed=0, generic_names=("ABIO_DIC14",), pairs=(("OCMIP_BGC_PO4", "OCMIP_BGC_DOP"), ("DIC_SHADOW", "ALK_SHADOW")),
                        shallow_levels=3):
    """Fields gen_A looks up in tracer_fname."""
    rng = np.random.default_rng(seed + 11)
    km, (jmt, imt) = len(F["dz"]), F["KMT"].shape
    T = OrderedDict()
    T["SINK_RATE"] = np.abs(_smooth3(rng, km, jmt, imt)) * 2.0                     # 1/yr
    T["PV"] = 3.0e-3 * (0.5 + np.abs(_smooth2d(rng.standard_normal((jmt, imt)), 1)))   # cm/s
    T["D_SF"] = -1.0e-3 * np.abs(_smooth2d(rng.standard_normal((jmt, imt)), 1))
    for nm in generic_names:
        T[f"d_J_{nm}_d_{nm}"] = -1.0e-9 * np.abs(_smooth3(rng, km, jmt, imt))
        for k2 in range(shallow_levels):
            if k2 == 1:
                continue                                                             # a gap: "does not exist" branch
            T[f"d_J_{nm}_d_{nm}_k_{k2 + 1:02d}"] = 2.0e-10 * _smooth3(rng, km, jmt, imt)
    for (a, b) in pairs:
        T[f"d_J_{a}_d_{b}"] = 1.0e-8 * np.abs(_smooth3(rng, km, jmt, imt))
        T[f"d_J_{b}_d_{a}"] = 3.0e-9 * np.abs(_smooth3(rng, km, jmt, imt))
        T[f"d_SF_{a}_d_{b}"] = 1.0e-4 * _smooth2d(rng.standard_normal((jmt, imt)), 1)
    return T


def write_tracer_source_file(path, F, T, version=2):
    km, (jmt, imt) = len(F["dz"]), F["KMT"].shape
    dims = OrderedDict([("nlon", imt), ("nlat", jmt), ("z_t", km)])
    variables = [(nm, ["z_t", "nlat", "nlon"] if a.ndim == 3 else ["nlat", "nlon"], a.astype(np.float64), None) for nm, a in T.items()]
    nc3.write(path, dims, variables, version=version)
