"""Synthetic ocean-grid Jacobian generator (host tooling: tests, bench input, experiments).

Stands in for the reference's gen_A on machines that have no POP/CESM history files: it emits
a matrix with the *same on-disk schema* (reference src/grid.c:217-316, src/matrix.c:263-369,
3844-3939) and the *same stencil shapes and coefficient formulas* as the reference's
const/centred/donor/upwind3 assembly paths:

  * flat ordering j outer, i middle, k inner, ocean cells only (src/matrix.c:239-251)
  * tracer-major rows for coupled tracers (src/matrix.c:778-784)
  * periodic in i, land rows at j=0 and j=jmt-1 (src/matrix.c:795-798, 176-189)
  * advection centred/donor weights (src/matrix.c:1239-1273, 1320-1354, 1401-1435),
    upwind3 = QUICK weights 0.75/0.375/-0.125 with the 0.625 land fallback
    (src/matrix.c:1610-1690), diagonal = -sum(off-diagonals) (adv_enforce_divfree, :2195-2196)
  * hmix const: ah*HTE/HUS/TAREA*dt (src/matrix.c:2656-2678); "isop" adds 8 k+-1 x E/W/N/S
    cross terms on the pattern of src/matrix.c:881-930 (the K13/K23 terms of the Redi tensor) and, by default
    (isop_k33), the matching vertical term K33 = K11 s^2 that the model carries in its implicit vertical mixing --
    without it the tensor (K11, K13; K13, 0) is indefinite, i.e. anti-diffusive across the neutral surface, which no
    ocean model produces; isop_k33=False gives that round-1 recipe back
  * vmix: vdc/(0.5(dz_k+dz_k+-1))/dz_k*dt (src/matrix.c:2979-2988) with a mixed layer
  * sink const_shallow: -year_cnt*rate where z_t < depth (src/matrix.c:3085-3091)
  * duplicates summed, exact zeros stripped, rows sorted by column (src/matrix.c:3826-3832)

This is NOT the solver and never runs inside it.  Everything is vectorised numpy so that a
1 degree x 60 level matrix (n ~ 4e6, nnz ~ 8e7) is built in well under a minute.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field

import numpy as np

from . import nc3

FILL_DOUBLE = 9.969209968386869e36   # netCDF default _FillValue for NC_DOUBLE


@dataclass
class SynthProblem:
    imt: int
    jmt: int
    km: int
    coupled_tracer_cnt: int
    tracer_state_len: int
    KMT: np.ndarray                 # [jmt][imt] int32
    z_t: np.ndarray                 # [km] cm
    dz: np.ndarray                  # [km] cm
    TLONG: np.ndarray
    TLAT: np.ndarray
    int3_to_tracer_state_ind: np.ndarray   # [km][jmt][imt] int32, -1 on land
    ind_i: np.ndarray
    ind_j: np.ndarray
    ind_k: np.ndarray
    rowptr: np.ndarray              # [n+1] int32
    colind: np.ndarray              # [nnz] int32
    nzval: np.ndarray               # [nnz] float64
    meta: dict = field(default_factory=dict)

    @property
    def flat_len(self):
        return self.coupled_tracer_cnt * self.tracer_state_len

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def col_start(self):
        """Water-column boundaries of one tracer's state vector: rows where k == 0."""
        starts = np.flatnonzero(self.ind_k == 0).astype(np.int32)
        return np.concatenate([starts, np.array([self.tracer_state_len], np.int32)])

    def scipy_csr(self):
        import scipy.sparse as sp
        n = self.flat_len
        return sp.csr_matrix((self.nzval, self.colind, self.rowptr), shape=(n, n))


def pop_like_dz(km):
    """POP-like stretched layer thicknesses (cm): 10 m near the surface growing to 250 m."""
    k = np.arange(km, dtype=np.float64)
    ramp = 0.5 * (1.0 + np.tanh((k - 0.55 * km) / (0.12 * km)))
    dz = 1000.0 * (1.0 + 24.0 * ramp)
    return np.round(dz)


def _smooth2d(f, passes):
    for _ in range(passes):
        f = 0.25 * (np.roll(f, 1, 1) + np.roll(f, -1, 1)) + 0.5 * f
        g = f.copy()
        g[1:-1] = 0.25 * (f[:-2] + f[2:]) + 0.5 * f[1:-1]
        f = g
    return f


def make_bathymetry(imt, jmt, km, dz, seed=0, land_frac_blobs=6, min_levels=3):
    rng = np.random.default_rng(seed)
    lon = (np.arange(imt) + 0.5) * 360.0 / imt
    lat = -79.0 + (np.arange(jmt) + 0.5) * (168.0 / jmt)
    LON, LAT = np.meshgrid(lon, lat)
    land = np.zeros((jmt, imt), bool)
    for _ in range(land_frac_blobs):
        clon, clat = rng.uniform(0, 360), rng.uniform(-55, 65)
        a, b = rng.uniform(18, 45), rng.uniform(15, 40)
        dlon = (LON - clon + 180.0) % 360.0 - 180.0
        land |= (dlon / a) ** 2 + ((LAT - clat) / b) ** 2 < 1.0
    zbot = np.cumsum(dz)
    depth = _smooth2d(rng.uniform(0.15, 1.0, (jmt, imt)), max(2, imt // 40)) * 1.15
    depth = np.clip((depth - depth.min()) / (depth.max() - depth.min()), 0.05, 1.0) * zbot[-1]
    # continental shelves: shallower next to land
    shelf = _smooth2d(land.astype(np.float64), max(1, imt // 80))
    depth *= np.clip(1.0 - 1.5 * shelf, 0.08, 1.0)
    KMT = np.searchsorted(zbot, depth, side="left") + 1
    KMT = np.clip(KMT, min_levels, km).astype(np.int32)
    KMT[land] = 0
    KMT[0, :] = 0
    KMT[-1, :] = 0
    return KMT, lon, lat


def _shift(F, di=0, dj=0, dk=0, fill=0):
    """Value of F at (k+dk, j+dj, i+di); periodic in i, `fill` outside j/k range."""
    G = F
    if di:
        G = np.roll(G, -di, axis=2)
    if dj:
        H = np.full_like(G, fill)
        if dj > 0:
            H[:, :-dj, :] = G[:, dj:, :]
        else:
            H[:, -dj:, :] = G[:, :dj, :]
        G = H
    if dk:
        H = np.full_like(G, fill)
        if dk > 0:
            H[:-dk] = G[dk:]
        else:
            H[-dk:] = G[:dk]
        G = H
    return G


# slot table: (di, dj, dk) in the reference's insertion order (src/matrix.c:800-930)
_SLOTS7 = [(0, 0, 0), (0, 0, -1), (0, 0, 1), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)]
_SLOTS_UW3 = [(0, 0, -2), (0, 0, 2), (2, 0, 0), (-2, 0, 0), (0, 2, 0), (0, -2, 0)]
_SLOTS_ISOP = [(1, 0, -1), (1, 0, 1), (-1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, 1, 1), (0, -1, -1), (0, -1, 1)]


def generate(imt=12, jmt=10, km=6, seed=0, adv="centred", hmix="const", coupled_tracer_cnt=1,
             day_cnt=365.0, u_scale=3.0, noise=0.3, ah=4.0e6, vdc_bg=0.1, vdc_ml=1000.0,
             sink_rate=365.0, sink_depth=10.0e2, min_cos=0.3, isop_k33=True):
    """Build a SynthProblem.  adv in {none, donor, centred, upwind3}; hmix in {const, isop}."""
    rng = np.random.default_rng(seed + 1000)
    dz = pop_like_dz(km)
    z_t = np.cumsum(dz) - 0.5 * dz
    KMT, lon, lat = make_bathymetry(imt, jmt, km, dz, seed)
    delta_t = 86400.0 * day_cnt
    year_cnt = day_cnt / 365.0

    R = 6.37122e8
    dlam = np.deg2rad(360.0 / imt)
    dphi = np.deg2rad(168.0 / jmt)
    # POP's displaced-pole grids have no polar singularity: zonal spacing never collapses the way a
    # regular lat-lon grid's does, so the metric floors cos(lat) at min_cos
    coslat = np.maximum(np.cos(np.deg2rad(lat)), min_cos)
    dx = (R * coslat * dlam)[None, :, None]                      # T-cell width        [1,j,1]
    dy = np.full((1, jmt, 1), R * dphi)
    latn = np.deg2rad(lat + 0.5 * 168.0 / jmt)
    dxn = (R * np.maximum(np.cos(latn), min_cos) * dlam)[None, :, None]   # north-face length
    TAREA = dx * dy

    kk = np.arange(km)[:, None, None]
    M = kk < KMT[None]                                            # ocean mask [k,j,i]
    # index maps: j outer, i middle, k inner
    Mt = np.transpose(M, (1, 2, 0))
    tsl = int(Mt.sum())
    IDXt = np.full(Mt.shape, -1, np.int32)
    IDXt[Mt] = np.arange(tsl, dtype=np.int32)
    IDX = np.ascontiguousarray(np.transpose(IDXt, (2, 0, 1)))
    jj, ii, kk3 = np.nonzero(Mt)
    ind_i, ind_j, ind_k = ii.astype(np.int32), jj.astype(np.int32), kk3.astype(np.int32)

    def exists(di, dj, dk):
        return M & _shift(M, di, dj, dk, False)

    # ------------------------------------------------------------------ circulation
    # corner (U-point) streamfunction per level: gyres + eddy noise, masked so that no
    # transport crosses a face that touches land (KMU-style mask)
    KMU = np.minimum(np.minimum(KMT, np.roll(KMT, -1, 1)),
                     np.minimum(np.vstack([KMT[1:], KMT[-1:]]), np.roll(np.vstack([KMT[1:], KMT[-1:]]), -1, 1)))
    MU = kk < KMU[None]
    LONc, LATc = np.meshgrid(np.deg2rad(lon), np.deg2rad(lat))
    gyre = np.sin(3.0 * LATc) * np.sin(2.0 * LONc) + 0.5 * np.sin(5.0 * LATc + 1.0) * np.cos(3.0 * LONc)
    eddy = _smooth2d(rng.standard_normal((jmt, imt)), 1)
    eddy /= np.abs(eddy).max()
    vert = np.exp(-z_t / 80000.0)[:, None, None] + 0.1                                # surface intensified
    L = R * dphi
    psi = u_scale * (0.3 * R * gyre[None] + noise * L * eddy[None]) * vert * MU        # cm^2/s
    UTE = -(psi - _shift(psi, 0, -1, 0))                                               # through east face
    VTN = psi - _shift(psi, -1, 0, 0)                                                  # through north face
    # meridional overturning potential on (top edge of north face): gives VTN and WVEL parts
    both = np.minimum(KMT, np.vstack([KMT[1:], KMT[-1:]]))                             # min(KMT[j],KMT[j+1])
    MPhi = (kk >= 1) & (kk < both[None])
    zi = (np.cumsum(dz) - dz)[:, None, None]                                           # depth of top interface
    Phi = 2.0e-3 * u_scale * TAREA * np.sin(np.pi * np.clip(zi / 3.0e5, 0, 1)) * np.cos(2.0 * LATc)[None] * MPhi
    Phi_below = _shift(Phi, 0, 0, 1)
    # transport through north face per unit depth (cm^2/s): Phi is a volume flux (cm^3/s)
    VTN = VTN + (Phi - Phi_below) / dz[:, None, None]
    WVEL = -(Phi - _shift(Phi, 0, -1, 0)) / TAREA                                      # top face, + up
    WVEL[0] = 0.0

    nslots = 7 + (6 if adv == "upwind3" else 0) + (8 if hmix == "isop" else 0)
    slots = list(_SLOTS7) + (list(_SLOTS_UW3) if adv == "upwind3" else []) + (list(_SLOTS_ISOP) if hmix == "isop" else [])
    V = np.zeros((nslots,) + M.shape)
    S = {s: n for n, s in enumerate(slots)}
    E1, W1, N1, S1 = exists(1, 0, 0), exists(-1, 0, 0), exists(0, 1, 0), exists(0, -1, 0)
    UP, DN = exists(0, 0, -1), exists(0, 0, 1)
    Uw, Vs = _shift(UTE, -1, 0, 0), _shift(VTN, 0, -1, 0)
    Wb = _shift(WVEL, 0, 0, 1)
    cA = delta_t / TAREA
    cZ = delta_t / dz[:, None, None]

    if adv in ("centred", "donor"):
        def w(cond):
            return 0.5 if adv == "centred" else cond.astype(np.float64)
        we, ww = w(UTE > 0), w(Uw < 0)
        wn, ws = w(VTN > 0), w(Vs < 0)
        wt, wb = w(WVEL > 0), w(Wb < 0)
        V[S[(1, 0, 0)]] -= (1 - we) * UTE * cA * E1
        V[S[(-1, 0, 0)]] += (1 - ww) * Uw * cA * W1
        V[S[(0, 1, 0)]] -= (1 - wn) * VTN * cA * N1
        V[S[(0, -1, 0)]] += (1 - ws) * Vs * cA * S1
        V[S[(0, 0, -1)]] -= (1 - wt) * WVEL * cZ * UP
        V[S[(0, 0, 1)]] += (1 - wb) * Wb * cZ * DN
    elif adv == "upwind3":
        def quick(Tpos_out, Tneg_out, Tpos_in, Tneg_in, c, d):
            """Face pair along direction d=(di,dj,dk): *_out = far-side face, *_in = near-side."""
            p1, m1 = exists(*d), exists(-d[0], -d[1], -d[2])
            p2 = exists(2 * d[0], 2 * d[1], 2 * d[2])
            m2 = exists(-2 * d[0], -2 * d[1], -2 * d[2])
            f = lambda ok: np.where(ok, 0.75, 0.625)
            V[S[d]] += (-0.375 * Tpos_out - f(p2) * Tneg_out - 0.125 * Tneg_in) * c * p1
            V[S[(-d[0], -d[1], -d[2])]] += (0.125 * Tpos_out + f(m2) * Tpos_in + 0.375 * Tneg_in) * c * m1
            V[S[(2 * d[0], 2 * d[1], 2 * d[2])]] += 0.125 * Tneg_out * c * p2
            V[S[(-2 * d[0], -2 * d[1], -2 * d[2])]] += -0.125 * Tpos_in * c * m2
        pos, neg = (lambda T: np.maximum(T, 0.0)), (lambda T: np.minimum(T, 0.0))
        quick(pos(UTE), neg(UTE), pos(Uw), neg(Uw), cA, (1, 0, 0))
        quick(pos(VTN), neg(VTN), pos(Vs), neg(Vs), cA, (0, 1, 0))
        # vertical: "out" face is the top face (toward k-1), flow + upward
        quick(pos(WVEL), neg(WVEL), pos(Wb), neg(Wb), cZ, (0, 0, -1))
    if adv != "none":
        V[0] = -V[1:].sum(axis=0)       # adv_enforce_divfree: diag = -sum(off-diag)

    # ------------------------------------------------------------------ horizontal mixing
    ce = ah * dy / dx * cA * E1
    cw = ah * dy / dx * cA * W1
    cn = ah * dxn / dy * cA * N1
    dxs = np.concatenate([dxn[:, :1], dxn[:, :-1]], axis=1)
    cs = ah * dxs / dy * cA * S1
    V[0] -= ce + cw + cn + cs
    V[S[(1, 0, 0)]] += ce
    V[S[(-1, 0, 0)]] += cw
    V[S[(0, 1, 0)]] += cn
    V[S[(0, -1, 0)]] += cs
    if hmix == "isop":
        taper = np.clip(z_t / 3.0e4, 0, 1)[:, None, None]
        sx = _smooth2d(rng.standard_normal((jmt, imt)), 2)[None] * taper
        sy = _smooth2d(rng.standard_normal((jmt, imt)), 2)[None] * taper
        sx, sy = 1.5 * sx / np.abs(sx).max(), 1.5 * sy / np.abs(sy).max()
        for (d, cface, s) in (((1, 0, 0), ce, sx), ((-1, 0, 0), cw, -_shift(sx, -1, 0, 0)),
                              ((0, 1, 0), cn, sy), ((0, -1, 0), cs, -_shift(sy, 0, -1, 0))):
            ok = UP & DN & exists(d[0], d[1], -1) & exists(d[0], d[1], 1) & exists(*d)
            x = 0.25 * cface * s * ok
            V[S[(d[0], d[1], -1)]] += x
            V[S[(d[0], d[1], 1)]] -= x
            V[S[(0, 0, -1)]] += x
            V[S[(0, 0, 1)]] -= x
        if isop_k33:
            # vertical part K33 = K11 s^2 of the Redi tensor: without it the (K11, K13) terms above form an indefinite
            # tensor (anti-diffusive across the neutral surface); the model carries K33 in its implicit vertical mixing
            kz = 0.5 * (ce * sx ** 2 + cw * _shift(sx, -1, 0, 0) ** 2 + cn * sy ** 2 + cs * _shift(sy, 0, -1, 0) ** 2)
            k33_top = 0.5 * (kz + _shift(kz, 0, 0, -1)) * UP
            k33_bot = _shift(0.5 * (kz + _shift(kz, 0, 0, -1)), 0, 0, 1) * DN
            V[0] -= k33_top + k33_bot
            V[S[(0, 0, -1)]] += k33_top
            V[S[(0, 0, 1)]] += k33_bot

    # ------------------------------------------------------------------ vertical mixing
    mld = (3000.0 + 27000.0 * np.abs(np.sin(np.deg2rad(lat))) ** 3)[None, :, None]   # cm
    ztop = (np.cumsum(dz) - dz)[:, None, None]
    vdc_top = vdc_bg + vdc_ml * (ztop < mld)                                           # at top interface of cell k
    dzt = np.empty(km)
    dzt[0] = dz[0]
    dzt[1:] = 0.5 * (dz[:-1] + dz[1:])
    ct = vdc_top / dzt[:, None, None] * cZ * UP
    cb = _shift(vdc_top / dzt[:, None, None], 0, 0, 1) * cZ * DN
    V[0] -= ct + cb
    V[S[(0, 0, -1)]] += ct
    V[S[(0, 0, 1)]] += cb

    # ------------------------------------------------------------------ sinks
    V[0] += (-year_cnt * sink_rate) * (z_t < sink_depth)[:, None, None]

    # ------------------------------------------------------------------ assemble CSR
    cols = np.empty((nslots,) + M.shape, np.int64)
    for n, (di, dj, dk) in enumerate(slots):
        c = _shift(IDX, di, dj, dk, -1).astype(np.int64)
        if n > 0:
            c[~exists(di, dj, dk)] = -1
        cols[n] = c
    sel = np.transpose(M, (1, 2, 0))                                                   # (j,i,k) order
    Vr = np.transpose(V, (2, 3, 1, 0))[sel]                                            # [tsl, nslots]
    Cr = np.transpose(cols, (2, 3, 1, 0))[sel]
    del V, cols
    # sum duplicates (periodic wrap on tiny grids can alias E/W or E2/W2): src/matrix.c:3620-3650
    order = np.argsort(Cr, axis=1, kind="stable")
    Cr = np.take_along_axis(Cr, order, 1)
    Vr = np.take_along_axis(Vr, order, 1)
    for s in range(nslots - 1, 0, -1):
        dup = (Cr[:, s] == Cr[:, s - 1]) & (Cr[:, s] >= 0)
        if dup.any():
            Vr[dup, s - 1] += Vr[dup, s]
            Vr[dup, s] = 0.0
            Cr[dup, s] = -1

    cnt = coupled_tracer_cnt
    blocks_c, blocks_v = [], []
    for t in range(cnt):
        Vt = Vr.copy() if cnt > 1 else Vr
        Ct = np.where(Cr >= 0, Cr + t * tsl, -1)
        if cnt > 1:
            # per-tracer extra decay so the diagonal blocks differ (sink_const, src/matrix.c:3072-3081)
            diag_slot = (Cr == np.arange(tsl)[:, None])
            Vt[diag_slot] += -year_cnt * 0.05 * t
            # same-cell coupling entries to the other tracers (src/matrix.c:955-961, 3274-3383)
            depth_w = np.exp(-z_t[ind_k] / 3.0e4)
            extra_c = np.empty((tsl, cnt - 1), np.int64)
            extra_v = np.empty((tsl, cnt - 1))
            m = 0
            for t2 in range(cnt):
                if t2 == t:
                    continue
                rate = year_cnt * (2.0 if (t2 == (t - 1) % cnt) else 0.25) * depth_w
                extra_c[:, m] = t2 * tsl + np.arange(tsl)
                extra_v[:, m] = rate
                Vt[diag_slot] -= rate
                m += 1
            Ct = np.concatenate([Ct, extra_c], axis=1)
            Vt = np.concatenate([Vt, extra_v], axis=1)
        blocks_c.append(Ct)
        blocks_v.append(Vt)
    C = np.concatenate(blocks_c, 0) if cnt > 1 else blocks_c[0]
    Vv = np.concatenate(blocks_v, 0) if cnt > 1 else blocks_v[0]
    keep = (C >= 0) & (Vv != 0.0)                                                      # strip zeros: :3656-3688
    big = np.iinfo(np.int64).max
    Cs = np.where(keep, C, big)
    order = np.argsort(Cs, axis=1, kind="stable")                                      # sort cols: :3731-3770
    Cs = np.take_along_axis(Cs, order, 1)
    Vs_ = np.take_along_axis(Vv, order, 1)
    keep = Cs != big
    rowlen = keep.sum(axis=1)
    rowptr = np.zeros(cnt * tsl + 1, np.int64)
    np.cumsum(rowlen, out=rowptr[1:])
    if rowptr[-1] >= 2 ** 31:
        raise ValueError("nnz exceeds the int32 limit of the file schema")
    colind = Cs[keep].astype(np.int32)
    nzval = np.ascontiguousarray(Vs_[keep])
    TLONG = np.broadcast_to(lon[None, :], (jmt, imt)).copy()
    TLAT = np.broadcast_to(lat[:, None], (jmt, imt)).copy()
    return SynthProblem(imt, jmt, km, cnt, tsl, KMT, z_t, dz, TLONG, TLAT, IDX, ind_i, ind_j, ind_k,
                        rowptr.astype(np.int32), colind, nzval,
                        meta=dict(seed=seed, adv=adv, hmix=hmix, day_cnt=day_cnt, u_scale=u_scale, noise=noise,
                                  ah=ah, vdc_bg=vdc_bg, vdc_ml=vdc_ml, sink_rate=sink_rate, sink_depth=sink_depth,
                                  min_cos=min_cos, isop_k33=isop_k33))


def write_matrix_file(p: SynthProblem, path, version=2):
    """Emit the matrix file with the schema of SURVEY.md section 3.3 (what gen_A writes)."""
    dims = OrderedDict([("nlon", p.imt), ("nlat", p.jmt), ("z_t", p.km),
                        ("tracer_state_len", p.tracer_state_len), ("nnz", p.nnz),
                        ("flat_len_p1", p.flat_len + 1)])
    m1 = np.int32(-1)
    variables = [
        ("z_t", ["z_t"], p.z_t.astype(np.float64),
         OrderedDict(long_name="depth from surface to midpoint of layer", units="centimeters", positive="down")),
        ("TLONG", ["nlat", "nlon"], p.TLONG, OrderedDict(long_name="array of t-grid longitudes", units="degrees_east")),
        ("TLAT", ["nlat", "nlon"], p.TLAT, OrderedDict(long_name="array of t-grid latitudes", units="degrees_north")),
        ("KMT", ["nlat", "nlon"], p.KMT.astype(np.int32),
         OrderedDict(long_name="k Index of Deepest Grid Cell on T Grid", coordinates="TLONG TLAT")),
        ("int3_to_tracer_state_ind", ["z_t", "nlat", "nlon"], p.int3_to_tracer_state_ind.astype(np.int32),
         OrderedDict([("coordinates", "TLONG TLAT"), ("_FillValue", m1), ("missing_value", m1)])),
        ("tracer_state_ind_to_i", ["tracer_state_len"], p.ind_i, None),
        ("tracer_state_ind_to_j", ["tracer_state_len"], p.ind_j, None),
        ("tracer_state_ind_to_k", ["tracer_state_len"], p.ind_k, None),
        ("coupled_tracer_cnt", [], np.array(p.coupled_tracer_cnt, np.int32), None),
        ("nzval_row_wise", ["nnz"], p.nzval, None),
        ("colind", ["nnz"], p.colind, None),
        ("rowptr", ["flat_len_p1"], p.rowptr, None),
    ]
    nc3.write(path, dims, variables, version=version)


def make_tracer_fields(p: SynthProblem, names, seed=1, with_time_dim=False):
    """Standard-normal RHS on ocean cells, netCDF fill value on land (must survive the solve)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    ocean = p.int3_to_tracer_state_ind >= 0
    for nm in names:
        f = np.full((p.km, p.jmt, p.imt), FILL_DOUBLE)
        f[ocean] = rng.standard_normal(int(ocean.sum()))
        out[nm] = f
    return out


def write_tracer_file(p: SynthProblem, path, fields, version=2, nc_type="float64"):
    dims = OrderedDict([("nlon", p.imt), ("nlat", p.jmt), ("z_t", p.km)])
    variables = [(nm, ["z_t", "nlat", "nlon"], f.astype(nc_type),
                  OrderedDict([("_FillValue", np.asarray(FILL_DOUBLE).astype(nc_type)[()])]))
                 for nm, f in fields.items()]
    nc3.write(path, dims, variables, version=version)


def flatten(p: SynthProblem, fields_list):
    """B[t*tsl + s] = field_t[k_s][j_s][i_s]  (reference src/solve_ABglobal.c:184-191)."""
    return np.concatenate([f[p.ind_k, p.ind_j, p.ind_i] for f in fields_list])


def tracer_rows(p1: SynthProblem, t, cnt):
    """Rows of tracer `t` of the `cnt`-tracer coupled problem built on the single-tracer problem `p1`
    (same recipe as generate(coupled_tracer_cnt=cnt): per-tracer decay on the diagonal, same-cell
    coupling entries to every other tracer) -- without ever forming the other tracers' rows.

    Returns (rowptr_local, colind_global, val): what one rank of a tracer-per-rank partition owns
    (reference src/matrix.c:778-784: rows are tracer-major, so tracer t is the contiguous row block
    [t * tracer_state_len, (t + 1) * tracer_state_len)).
    """
    if p1.coupled_tracer_cnt != 1:
        raise ValueError("p1 must be a single-tracer problem")
    tsl = p1.tracer_state_len
    if cnt == 1:
        return p1.rowptr.copy(), p1.colind.copy(), p1.nzval.copy()
    year_cnt = p1.meta["day_cnt"] / 365.0
    rp = p1.rowptr.astype(np.int64)
    length = np.diff(rp)
    rows = np.repeat(np.arange(tsl, dtype=np.int64), length)
    v = p1.nzval.copy()
    diag = np.flatnonzero(p1.colind == rows)
    if diag.size != tsl:
        raise ValueError("single-tracer matrix lacks a diagonal entry")
    v[diag] += -year_cnt * 0.05 * t
    depth_w = np.exp(-p1.z_t[p1.ind_k] / 3.0e4)
    others = [t2 for t2 in range(cnt) if t2 != t]
    rates = {}
    for t2 in others:
        rates[t2] = year_cnt * (2.0 if (t2 == (t - 1) % cnt) else 0.25) * depth_w
        v[diag] -= rates[t2]
    new_rp = np.zeros(tsl + 1, np.int64)
    np.cumsum(length + (cnt - 1), out=new_rp[1:])
    ci = np.empty(new_rp[-1], np.int32)
    val = np.empty(new_rp[-1], np.float64)
    cell = np.arange(tsl, dtype=np.int64)
    for m, t2 in enumerate(others):                      # t2 < t sorts before the tracer's own columns, t2 > t after
        pos = new_rp[:-1] + (m if t2 < t else m + length)
        ci[pos] = t2 * tsl + cell
        val[pos] = rates[t2]
    own = new_rp[:-1][rows] + t + (np.arange(rp[-1], dtype=np.int64) - rp[:-1][rows])
    ci[own] = p1.colind.astype(np.int64) + t * tsl
    val[own] = v
    return new_rp.astype(np.int32), ci, val
