"""CPU restatement (numpy) of the reference's matrix generator gen_A.

TEST INFRASTRUCTURE ONLY: imported by tests/ to check host/matrix_gen.c + cli/gen_A_main.c;
never imported by the product.

PARITY UNPINNED: the reference ships no input files, golden matrices or assertions for gen_A
(test/test_gen_A.csh only checks the exit status against files that are not in the repository)
and it cannot be built here (src/matrix.c:74 includes SuperLU_DIST's superlu_ddefs.h, I/O is
libnetcdf; neither is installed).  This module is a second, independently structured reading
of the same source text: where host/matrix_gen.c walks rows and addresses named slots, this
one holds the whole pattern as [rows x slots] arrays and applies each term as a masked
array expression.  Agreement of the two to the last bit is what the tests assert.

Followed text (reference src/):
  grid.c:139-202        KMT clean-up, region mask, polar-row check, KMU
  matrix.c:210-259      index maps (j outer, i middle, k inner)
  matrix.c:596-662, 753-981   pattern order inside a row
  matrix.c:986-1451     centred / donor advection     :1455-2017  upwind3
  matrix.c:2094-2207    adv_enforce_divfree           :2211-2387  isopycnal IRF mixing
  matrix.c:2391-2726    hor_file / const lateral mixing
  matrix.c:2776-3014    vertical mixing (matrix_file, file, const)
  matrix.c:3059-3617    sinks, generic tracer, coupled pairs, pv, d_SF
  matrix.c:3621-3770    duplicate folding, zero stripping, column sort
  matrix.c:3775-3840    order of the passes
"""
from __future__ import annotations

import numpy as np

S7 = [(0, 0, 0), (0, 0, -1), (0, 0, 1), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)]            # (di, dj, dk)
S_UW3 = [(0, 0, -2), (0, 0, 2), (2, 0, 0), (-2, 0, 0), (0, 2, 0), (0, -2, 0)]
S_ISOP = [(1, 0, -1), (1, 0, 1), (-1, 0, -1), (-1, 0, 1), (0, 1, -1), (0, 1, 1), (0, -1, -1), (0, -1, 1)]
PAIR_NAMES = {"OCMIP_BGC_PO4_DOP": ("OCMIP_BGC_PO4", "OCMIP_BGC_DOP"), "DIC_SHADOW_ALK_SHADOW": ("DIC_SHADOW", "ALK_SHADOW")}


def default_options():
    """gen_A.c:95-111, 67-92."""
    return dict(day_cnt=365.0, adv="cent", divfree=True, hmix="isop_file", vmix="file", coupled_tracer_cnt=1,
                coupled_type="none", per_tracer=[dict(sink=("none",), pv=None, sf=None)])


def masked_grid(F, reg=None):
    KMT = np.array(F["KMT"], dtype=np.int64)
    KMT[KMT < 0] = 0
    if reg is not None:
        inner = KMT[1:-1]
        inner[np.asarray(reg)[1:-1] < 0] = 0
    if KMT[0].any() or KMT[-1].any():
        raise ValueError("non-land found on a polar row")
    north = np.vstack([KMT[1:], KMT[-1:]])
    KMU = np.minimum(np.minimum(KMT, north), np.minimum(np.roll(KMT, -1, 1), np.roll(north, -1, 1)))
    KMU[-1] = 0
    return KMT, KMU


class _Gen:
    def __init__(self, F, fills, opts, T, reg):
        self.F, self.fills, self.o, self.T = F, fills, opts, T or {}
        self.KMT, self.KMU = masked_grid(F, reg)
        self.jmt, self.imt = self.KMT.shape
        self.km = len(F["dz"])
        self.dz = np.asarray(F["dz"], np.float64)
        self.z_t = np.asarray(F["z_t"], np.float64)
        self.TAREA = np.asarray(F["TAREA"], np.float64)
        self.delta_t = 60.0 * 60.0 * 24.0 * opts["day_cnt"]
        self.year_cnt = opts["day_cnt"] / 365.0
        km, jmt, imt = self.km, self.jmt, self.imt
        wet = np.arange(km)[None, None, :] < self.KMT[:, :, None]                 # [j, i, k]
        self.rj, self.ri, self.rk = np.nonzero(wet)
        self.tsl = len(self.rk)
        self.IDX = np.full((km, jmt, imt), -1, np.int64)
        self.IDX[self.rk, self.rj, self.ri] = np.arange(self.tsl)
        self.cnt = opts["coupled_tracer_cnt"]

    # ---- helpers ------------------------------------------------------------------------
    def nb(self, d):
        """(exists, column) of the neighbour at offset d = (di, dj, dk) for every row."""
        di, dj, dk = d
        ii, jj, kk = (self.ri + di) % self.imt, self.rj + dj, self.rk + dk
        ok = (jj >= 0) & (jj < self.jmt) & (kk >= 0) & (kk < self.km)
        jc, kc = np.clip(jj, 0, self.jmt - 1), np.clip(kk, 0, self.km - 1)
        ok &= kc < self.KMT[jc, ii]
        return ok, np.where(ok, self.IDX[kc, jc, ii], -1)

    def f3(self, name, fv=True, src=None):
        src = self.F if src is None else src
        a = np.array(src[name], dtype=np.float64).reshape(self.km, self.jmt, self.imt)
        if fv:
            a[a == np.float64(self.fills[name])] = 0.0
        return a

    def f2(self, name, fv=True, src=None):
        src = self.F if src is None else src
        a = np.array(src[name], dtype=np.float64).reshape(self.jmt, self.imt)
        if fv:
            a[a == np.float64(self.fills[name])] = 0.0
        return a

    def at(self, A, di=0, dj=0, dk=0):
        """A[k+dk, j+dj, i+di] per row (indices clipped; only used under an existence mask)."""
        ii = (self.ri + di) % self.imt
        jj = np.clip(self.rj + dj, 0, self.jmt - 1)
        if A.ndim == 2:
            return A[jj, ii]
        return A[np.clip(self.rk + dk, 0, self.km - 1), jj, ii]

    # ---- pattern ------------------------------------------------------------------------
    def build_pattern(self):
        o = self.o
        self.slots = list(S7) + (S_UW3 if o["adv"] == "upwind3" else []) + (S_ISOP if o["hmix"] == "isop_file" else [])
        self.ex, cols = {}, []
        for d in self.slots:
            ok, c = self.nb(d)
            self.ex[d] = ok
            cols.append(c)
        self.ns = len(self.slots)
        col_kmt = self.KMT[self.rj, self.ri]
        self.vm0 = len(cols)
        if o["vmix"] == "matrix_file":
            for k2 in range(self.km):
                cols.append(np.where(k2 < col_kmt, self.IDX[k2, self.rj, self.ri], -1))
        base = np.stack(cols, 1) if cols else np.zeros((self.tsl, 0), np.int64)
        self.C, self.V, self.sink0, self.other0 = [], [], [], []
        for t in range(self.cnt):
            extra = []
            sink = o["per_tracer"][t]["sink"]
            if sink[0] == "generic_tracer":
                kmax = self.km - 1 if sink[2] == -1 else sink[2] - 1
                start = np.minimum(self.rk, kmax)
                for n in range(kmax + 1):                       # n-th entry of the run is level start - n
                    k2 = start - n
                    extra.append(np.where(k2 >= 0, self.IDX[np.clip(k2, 0, None), self.rj, self.ri], -1))
            self.sink0.append(base.shape[1])
            self.other0.append(base.shape[1] + len(extra))
            own = np.where(base >= 0, base + t * self.tsl, -1)
            ex_own = [np.where(e >= 0, e + t * self.tsl, -1) for e in extra]
            others = [t2 * self.tsl + np.arange(self.tsl) for t2 in range(self.cnt) if t2 != t]
            Ct = np.concatenate([own] + [e[:, None] for e in ex_own + others], 1)
            self.C.append(Ct)
            self.V.append(np.zeros(Ct.shape))

    def sl(self, d):
        return self.slots.index(d)

    def sub(self, t, d, mask, expr):
        v = self.V[t][:, self.sl(d)]
        v[mask] -= expr[mask]

    def add(self, t, d, mask, expr):
        v = self.V[t][:, self.sl(d)]
        v[mask] += expr[mask]

    # ---- advection ----------------------------------------------------------------------
    def face_transports(self):
        hor = self.o["hmix"] == "hor_file"
        km, jmt, imt = self.km, self.jmt, self.imt
        kk = np.arange(km)[:, None, None]
        U, DY = self.f3("UVEL"), self.f2("DYU")
        UTE = np.zeros((km, jmt, imt))
        UTE[:, 1:-1] += np.where(kk < self.KMU[None, 1:-1], 0.5 * U[:, 1:-1] * DY[None, 1:-1], 0.0)
        UTE[:, 1:-1] += np.where(kk < self.KMU[None, :-2], 0.5 * U[:, :-2] * DY[None, :-2], 0.0)
        if hor:
            W, H = self.f3("UISOP", fv=False), self.f2("HTE")
            m = (kk < self.KMT[None]) & (kk < np.roll(self.KMT, -1, 1)[None])
            UTE[:, 1:-1] += np.where(m[:, 1:-1], W[:, 1:-1] * H[None, 1:-1], 0.0)
        Vv, DX = self.f3("VVEL"), self.f2("DXU")
        VTN = np.zeros((km, jmt, imt))
        VTN[:, 1:-1] += np.where(kk < self.KMU[None, 1:-1], 0.5 * Vv[:, 1:-1] * DX[None, 1:-1], 0.0)
        Vw, DXw, KMUw = np.roll(Vv, 1, 2), np.roll(DX, 1, 1), np.roll(self.KMU, 1, 1)
        VTN[:, 1:-1] += np.where(kk < KMUw[None, 1:-1], 0.5 * Vw[:, 1:-1] * DXw[None, 1:-1], 0.0)
        if hor:
            W, H = self.f3("VISOP"), self.f2("HTN")
            m = (kk[:, :, :] < self.KMT[None, 1:-1]) & (kk < self.KMT[None, 2:])
            VTN[:, 1:-1] += np.where(m, W[:, 1:-1] * H[None, 1:-1], 0.0)
        WV = np.zeros((km, jmt, imt))
        for nm in (["WVEL", "WISOP"] if hor else ["WVEL"]):
            W = self.f3(nm)
            WV[:, 1:-1] += np.where(kk < self.KMT[None, 1:-1], W[:, 1:-1], 0.0)
        WV[0, 1:-1] = 0.0
        return UTE, VTN, WV

    def adv_two_point(self):
        dt, TA = self.delta_t, self.at(self.TAREA)
        donor = self.o["adv"] == "donor"
        UTE, VTN, WV = self.face_transports()
        E, W, N, S, UP, DN = (self.ex[d] for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1)))
        dzk = self.dz[self.rk]
        for t in range(self.cnt):
            for (Fld, far, near, dfar, dnear, den, shift_near) in (
                    (UTE, E, W, (1, 0, 0), (-1, 0, 0), TA, dict(di=-1)),
                    (VTN, N, S, (0, 1, 0), (0, -1, 0), TA, dict(dj=-1))):
                out_, in_ = self.at(Fld), self.at(Fld, **shift_near)
                w_out = (out_ > 0.0).astype(np.float64) if donor else np.full(self.tsl, 0.5)
                w_in = (in_ < 0.0).astype(np.float64) if donor else np.full(self.tsl, 0.5)
                self.sub(t, (0, 0, 0), far, w_out * out_ / den * dt)
                self.add(t, (0, 0, 0), near, w_in * in_ / den * dt)
                self.sub(t, dfar, far, (1.0 - w_out) * out_ / den * dt)
                self.add(t, dnear, near, (1.0 - w_in) * in_ / den * dt)
            top, bot = self.at(WV), self.at(WV, dk=1)
            w_top = (top > 0.0).astype(np.float64) if donor else np.full(self.tsl, 0.5)
            w_bot = (bot < 0.0).astype(np.float64) if donor else np.full(self.tsl, 0.5)
            self.sub(t, (0, 0, 0), UP, w_top * top / dzk * dt)
            self.add(t, (0, 0, 0), DN, w_bot * bot / dzk * dt)
            self.sub(t, (0, 0, -1), UP, (1.0 - w_top) * top / dzk * dt)
            self.add(t, (0, 0, 1), DN, (1.0 - w_bot) * bot / dzk * dt)

    def adv_upwind3(self):
        dt, TA = self.delta_t, self.at(self.TAREA)
        ex = self.ex
        yes = np.ones(self.tsl, bool)
        for (pos, neg, ax) in (("UTE_POS", "UTE_NEG", (1, 0)), ("VTN_POS", "VTN_NEG", (0, 1))):
            P, Ng = self.f3(pos), self.f3(neg)
            p1, m1 = (ax[0], ax[1], 0), (-ax[0], -ax[1], 0)
            p2, m2 = (2 * ax[0], 2 * ax[1], 0), (-2 * ax[0], -2 * ax[1], 0)
            Po, No = self.at(P), self.at(Ng)                                     # far (east / north) face
            Pi, Ni = self.at(P, di=-ax[0], dj=-ax[1]), self.at(Ng, di=-ax[0], dj=-ax[1])   # near face
            for t in range(self.cnt):
                self.sub(t, (0, 0, 0), ex[m1], 0.75 * Po / TA * dt)
                self.sub(t, (0, 0, 0), ~ex[m1], (0.75 - 0.125) * Po / TA * dt)
                self.sub(t, (0, 0, 0), yes, 0.375 * No / TA * dt)
                self.add(t, (0, 0, 0), yes, 0.375 * Pi / TA * dt)
                self.add(t, (0, 0, 0), ex[p1], 0.75 * Ni / TA * dt)
                self.add(t, (0, 0, 0), ~ex[p1], (0.75 - 0.125) * Ni / TA * dt)
                self.sub(t, p1, ex[p1], 0.375 * Po / TA * dt)
                self.sub(t, p1, ex[p1] & ex[p2], 0.75 * No / TA * dt)
                self.sub(t, p1, ex[p1] & ~ex[p2], (0.75 - 0.125) * No / TA * dt)
                self.add(t, p1, ex[p1], (-0.125) * Ni / TA * dt)
                self.sub(t, m1, ex[m1], (-0.125) * Po / TA * dt)
                self.add(t, m1, ex[m1] & ex[m2], 0.75 * Pi / TA * dt)
                self.add(t, m1, ex[m1] & ~ex[m2], (0.75 - 0.125) * Pi / TA * dt)
                self.add(t, m1, ex[m1], 0.375 * Ni / TA * dt)
                self.sub(t, p2, ex[p2], (-0.125) * No / TA * dt)
                self.add(t, m2, ex[m2], (-0.125) * Pi / TA * dt)
        # vertical, stretched-grid weights (matrix.c:1868-1903)
        km, dz = self.km, self.dz
        dzc = np.concatenate([[dz[0]], dz, [dz[-1]]])                            # dzc[k] -> dzc[k + 1] here
        c = lambda k: dzc[k + 1]
        talfzp, tbetzp, tgamzp = np.zeros(km), np.zeros(km), np.zeros(km)
        talfzm, tbetzm, tdelzm = np.zeros(km), np.zeros(km), np.zeros(km)
        for k in range(km - 1):
            talfzp[k] = dz[k] * (2.0 * dz[k] + c(k - 1)) / (dz[k] + dz[k + 1]) / (c(k - 1) + 2.0 * dz[k] + dz[k + 1])
            tbetzp[k] = dz[k + 1] * (2.0 * dz[k] + c(k - 1)) / (dz[k] + dz[k + 1]) / (dz[k] + c(k - 1))
            tgamzp[k] = -(dz[k] * dz[k + 1]) / (dz[k] + c(k - 1)) / (dz[k + 1] + c(k - 1) + 2.0 * dz[k])
            talfzm[k] = dz[k] * (2.0 * dz[k + 1] + c(k + 2)) / (dz[k] + dz[k + 1]) / (dz[k + 1] + c(k + 2))
            tbetzm[k] = dz[k + 1] * (2.0 * dz[k + 1] + c(k + 2)) / (dz[k] + dz[k + 1]) / (dz[k] + c(k + 2) + 2.0 * dz[k + 1])
            tdelzm[k] = -(dz[k] * dz[k + 1]) / (dz[k + 1] + c(k + 2)) / (dz[k] + c(k + 2) + 2.0 * dz[k + 1])
        tbetzp[0] = tbetzp[0] + tgamzp[0]
        tgamzp[0] = 0.0
        P, Ng = self.f3("WTK_POS"), self.f3("WTK_NEG")
        P[0, 1:-1], Ng[0, 1:-1] = 0.0, 0.0
        Pt, Nt, Pb, Nb = self.at(P), self.at(Ng), self.at(P, dk=1), self.at(Ng, dk=1)
        k, km1 = self.rk, np.clip(self.rk - 1, 0, None)
        dzk = dz[k]
        UP, DN, UP2, DN2 = ex[(0, 0, -1)], ex[(0, 0, 1)], ex[(0, 0, -2)], ex[(0, 0, 2)]
        for t in range(self.cnt):
            self.sub(t, (0, 0, 0), UP & DN, talfzm[km1] * Pt / dzk * dt)
            self.sub(t, (0, 0, 0), UP & ~DN, (talfzm[km1] + tdelzm[km1]) * Pt / dzk * dt)
            self.sub(t, (0, 0, 0), UP, talfzp[km1] * Nt / dzk * dt)
            self.add(t, (0, 0, 0), DN, tbetzm[k] * Pb / dzk * dt)
            self.add(t, (0, 0, 0), DN, tbetzp[k] * Nb / dzk * dt)
            self.sub(t, (0, 0, -1), UP, tbetzm[km1] * Pt / dzk * dt)
            self.sub(t, (0, 0, -1), UP, tbetzp[km1] * Nt / dzk * dt)
            self.add(t, (0, 0, -1), UP & DN, tgamzp[k] * Nb / dzk * dt)
            self.sub(t, (0, 0, 1), DN & UP, tdelzm[km1] * Pt / dzk * dt)
            self.add(t, (0, 0, 1), DN & DN2, talfzm[k] * Pb / dzk * dt)
            self.add(t, (0, 0, 1), DN & ~DN2, (talfzm[k] + tdelzm[k]) * Pb / dzk * dt)
            self.add(t, (0, 0, 1), DN, talfzp[k] * Nb / dzk * dt)
            self.sub(t, (0, 0, -2), UP2, tgamzp[km1] * Nt / dzk * dt)
            self.add(t, (0, 0, 2), DN2, tdelzm[k] * Pb / dzk * dt)

    def divfree(self):
        adv_slots = [d for d in self.slots if d in S7[1:] or d in S_UW3]
        for t in range(self.cnt):
            s = np.zeros(self.tsl)
            for d in adv_slots:                                  # pattern order: 7-point ring, then second ring
                s = np.where(self.ex[d], s + self.V[t][:, self.sl(d)], s)
            self.V[t][:, 0] = -s

    # ---- mixing -------------------------------------------------------------------------
    def hmix_laplacian(self, kappa=None):
        dt, TA = self.delta_t, self.at(self.TAREA)
        HUS, HTE, HUW, HTN = (self.f2(n) for n in ("HUS", "HTE", "HUW", "HTN"))
        E, W, N, S = (self.ex[d] for d in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)))
        if kappa is None:
            ah = 4.0e6
            ce = ah * self.at(HTE) / self.at(HUS) / TA * dt
            cw = ah * self.at(HTE, di=-1) / self.at(HUS, di=-1) / TA * dt
            cn = ah * self.at(HTN) / self.at(HUW) / TA * dt
            cs = ah * self.at(HTN, dj=-1) / self.at(HUW, dj=-1) / TA * dt
        else:
            K = kappa
            ce = 0.5 * (self.at(K) + self.at(K, di=1)) * self.at(HTE) / self.at(HUS) / TA * dt
            cw = 0.5 * (self.at(K, di=-1) + self.at(K)) * self.at(HTE, di=-1) / self.at(HUS, di=-1) / TA * dt
            cn = 0.5 * (self.at(K) + self.at(K, dj=1)) * self.at(HTN) / self.at(HUW) / TA * dt
            cs = 0.5 * (self.at(K, dj=-1) + self.at(K)) * self.at(HTN, dj=-1) / self.at(HUW, dj=-1) / TA * dt
        ce, cw, cn, cs = np.where(E, ce, 0.0), np.where(W, cw, 0.0), np.where(N, cn, 0.0), np.where(S, cs, 0.0)
        yes = np.ones(self.tsl, bool)
        for t in range(self.cnt):
            self.sub(t, (0, 0, 0), yes, ce + cw + cn + cs)
            self.add(t, (1, 0, 0), E, ce)
            self.add(t, (-1, 0, 0), W, cw)
            self.add(t, (0, 1, 0), N, cn)
            self.add(t, (0, -1, 0), S, cs)

    def hmix_hor_file(self):
        if self.o["adv"] == "upwind3":
            raise ValueError("cannot use hmix_hor_file with adv_upwind3")
        K, H = self.f3("KAPPA_ISOP"), self.f3("HOR_DIFF")
        kk = np.arange(self.km)[:, None, None]
        K[:, 1:-1] += np.where(kk < self.KMT[None, 1:-1], H[:, 1:-1], 0.0)
        self.hmix_laplacian(K)

    def hmix_isop(self):
        dt = self.delta_t
        for d in S7 + S_ISOP:
            di, dj, dk = d
            ii, jj, kk = (self.ri + di) % self.imt, self.rj + dj, self.rk + dk
            ok = self.ex[d]
            val = np.zeros(self.tsl)
            for a in range(4):
                for b in range(3):
                    for c in range(3):
                        m = ok & (ii % 4 == a) & (jj % 3 == b) & (kk % 3 == c)
                        if not m.any():
                            continue
                        nm = f"HDIF_EXPLICIT_3D_IRF_{a + 1}_{b + 1}_{c + 1}"
                        if nm not in self.F:
                            nm = f"HDIF_EXPLICIT_3D_IRF_NK_{a + 1}_{b + 1}_{c + 1}"
                        val[m] = self.at(self.f3(nm, fv=False))[m] * dt
            for t in range(self.cnt):
                self.add(t, d, ok, val)

    def vmix(self, vdc3=None):
        dt, dz, k = self.delta_t, self.dz, self.rk
        UP, DN = self.ex[(0, 0, -1)], self.ex[(0, 0, 1)]
        km1, kp1 = np.clip(k - 1, 0, None), np.clip(k + 1, None, self.km - 1)
        top = 0.1 if vdc3 is None else self.at(vdc3, dk=-1)
        bot = 0.1 if vdc3 is None else self.at(vdc3)
        ct = np.where(UP, top / (0.5 * (dz[km1] + dz[k])) / dz[k] * dt, 0.0)
        cb = np.where(DN, bot / (0.5 * (dz[k] + dz[kp1])) / dz[k] * dt, 0.0)
        yes = np.ones(self.tsl, bool)
        for t in range(self.cnt):
            self.sub(t, (0, 0, 0), yes, ct + cb)
            self.add(t, (0, 0, -1), UP, ct)
            self.add(t, (0, 0, 1), DN, cb)

    def vmix_file(self):
        tot = self.f3("VDC_S")
        tot[:, 1:-1] += self.f3("VDC_GM")[:, 1:-1]
        self.vmix(tot)

    def vmix_matrix(self):
        kmt = self.KMT[self.rj, self.ri]
        for kp in range(self.km):
            G = self.at(self.f3(f"vmix_matrix_{kp + 1:03d}_CUR", fv=False)) * self.delta_t
            m = kp < kmt
            for t in range(self.cnt):
                v = self.V[t][:, self.vm0 + kp]
                v[m] += G[m]

    # ---- sinks / surface ----------------------------------------------------------------
    def sinks(self):
        o, T, dt, yc = self.o, self.T, self.delta_t, self.year_cnt
        yes = np.ones(self.tsl, bool)
        for t in range(self.cnt):
            sink = o["per_tracer"][t]["sink"]
            if sink[0] == "const":
                self.V[t][:, 0] += -yc * sink[1]
            elif sink[0] == "const_shallow":
                m = self.z_t[self.rk] < sink[2]
                self.V[t][m, 0] += -yc * sink[1]
            elif sink[0] == "file":
                self.add(t, (0, 0, 0), yes, -yc * self.at(self.f3(sink[1], fv=False, src=T)))
        for t in range(self.cnt):
            sink = o["per_tracer"][t]["sink"]
            if sink[0] != "generic_tracer":
                continue
            nm = sink[1]
            kmax = self.km - 1 if sink[2] == -1 else sink[2] - 1
            if f"d_J_{nm}_d_{nm}" in T:
                self.add(t, (0, 0, 0), yes, dt * self.at(self.f3(f"d_J_{nm}_d_{nm}", fv=False, src=T)))
            start = np.minimum(self.rk, kmax)
            for k2 in range(kmax + 1):
                name = f"d_J_{nm}_d_{nm}_k_{k2 + 1:02d}"
                if name not in T:
                    continue
                vals = dt * self.at(self.f3(name, fv=False, src=T))
                for n in range(kmax + 1):                         # the run entry that points at level k2
                    m = (start - n) == k2
                    v = self.V[t][:, self.sink0[t] + n]
                    v[m] += vals[m]

    def coupled(self, surface):
        typ = self.o["coupled_type"]
        if typ == "none" or (surface and typ != "DIC_SHADOW_ALK_SHADOW"):
            return
        names = PAIR_NAMES[typ]
        for t in range(self.cnt):
            for t2 in range(self.cnt):
                if t2 == t:
                    continue
                name = ("d_SF_%s_d_%s" if surface else "d_J_%s_d_%s") % (names[t], names[t2])
                if name not in self.T:
                    continue
                slot = self.other0[t] + (t2 if t2 < t else t2 - 1)
                v = self.V[t][:, slot]
                if surface:
                    m = self.rk == 0
                    v[m] += (self.delta_t * self.at(self.f2(name, fv=False, src=self.T)) / self.dz[0])[m]
                else:
                    v += self.delta_t * self.at(self.f3(name, fv=False, src=self.T))

    def surface_2d(self, key, sign):
        m = self.rk == 0
        for t in range(self.cnt):
            nm = self.o["per_tracer"][t].get(key)
            if nm is None:
                continue
            e = self.at(self.f2(nm, fv=False, src=self.T)) / self.dz[0] * self.delta_t
            v = self.V[t][:, 0]
            if sign < 0:
                v[m] -= e[m]
            else:
                v[m] += e[m]

    # ---- clean-up -----------------------------------------------------------------------
    def finish(self):
        width = max(c.shape[1] for c in self.C)        # tracers may carry different runs: pad with absent slots
        C = np.concatenate([np.pad(c, ((0, 0), (0, width - c.shape[1])), constant_values=-1) for c in self.C], 0)
        V = np.concatenate([np.pad(v, ((0, 0), (0, width - v.shape[1]))) for v in self.V], 0)
        ncol = C.shape[1]
        dup_cnt = 0
        for a in range(ncol):
            for b in range(a + 1, ncol):
                m = (C[:, a] >= 0) & (C[:, b] == C[:, a])
                if m.any():
                    V[m, a] += V[m, b]
                    V[m, b] = 0.0
                    dup_cnt += int(m.sum())
        keep = (C >= 0) & (V != 0.0)
        big = np.iinfo(np.int64).max
        Cs = np.where(keep, C, big)
        order = np.argsort(Cs, axis=1, kind="stable")
        Cs, Vs = np.take_along_axis(Cs, order, 1), np.take_along_axis(V, order, 1)
        keep = Cs != big
        rowptr = np.zeros(C.shape[0] + 1, np.int64)
        np.cumsum(keep.sum(1), out=rowptr[1:])
        return dict(rowptr=rowptr.astype(np.int32), colind=Cs[keep].astype(np.int32), nzval=np.ascontiguousarray(Vs[keep]),
                    nnz_pattern=int((C >= 0).sum()), dup_cnt=dup_cnt)


def gen_A(F, fills, opts, T=None, reg=None):
    """F: circulation-file variables as read back from the file; fills: name -> _FillValue."""
    g = _Gen(F, fills, opts, T, reg)
    g.build_pattern()
    adv = opts["adv"]
    if adv in ("cent", "donor"):
        g.adv_two_point()
    elif adv == "upwind3":
        g.adv_upwind3()
    if opts["divfree"]:
        g.divfree()
    {"none": lambda: None, "const": g.hmix_laplacian, "hor_file": g.hmix_hor_file, "isop_file": g.hmix_isop}[opts["hmix"]]()
    {"none": lambda: None, "const": g.vmix, "file": g.vmix_file, "matrix_file": g.vmix_matrix}[opts["vmix"]]()
    g.sinks()
    g.coupled(surface=False)
    g.surface_2d("pv", -1)
    g.surface_2d("sf", +1)
    g.coupled(surface=True)
    out = g.finish()
    out.update(KMT=g.KMT.astype(np.int32), int3_to_tracer_state_ind=g.IDX.astype(np.int32),
               ind_i=g.ri.astype(np.int32), ind_j=g.rj.astype(np.int32), ind_k=g.rk.astype(np.int32),
               tracer_state_len=g.tsl, flat_len=g.cnt * g.tsl)
    return out
