/* nkp_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library.  The product (libnkp_hip.so, the solve_AB* executables, the python package) never
 * links, imports or calls it, and has no CPU fallback of its own.
 *
 * PARITY UNPINNED by the reference: the reference repository holds no golden vectors, no
 * assertions and no runnable tests for this path, and the arithmetic it
 * delegates to -- SuperLU_DIST 5.1.3 (reference src/Makefile:3; calls at
 * src/solve_ABglobal.c:353,395 and src/solve_ABdist.c:518,571), ParMETIS and libnetcdf -- is
 * not vendored under /root/reference and not installed in this image, so the reference itself
 * cannot be built or run here.  What pins this oracle instead: fixtures under tests/golden/
 * generated in the build container by tests/golden/make_golden.py with SciPy 1.15.3's
 * scipy.sparse.linalg.splu -- the serial SuperLU of the same library family -- plus iterative
 * refinement, on the same CSR.  tests/test_oracle.py checks every oracle function against them.
 *
 * What is restated, and from where:
 *   ora_direct_solve   the contract of pdgssvx_ABglobal as the reference drives it
 *                      (set_default_options_dist: Equil=YES, IterRefine=SLU_DOUBLE;
 *                      src/solve_ABglobal.c:332-353, 363, 393-395): equilibrate, LU, solve,
 *                      refine in double until the componentwise backward error stagnates, B
 *                      overwritten by X, berr returned.  The factorisation itself is restated as
 *                      LAPACK-style banded LU with partial pivoting in the natural ordering
 *                      (SuperLU's supernodal GESP + ParMETIS ordering only changes fill, not x).
 *   ora_flatten / ora_unflatten      src/solve_ABglobal.c:184-191, 242-248
 *   ora_rowblock_partition           src/solve_ABdist.c:141-144
 *   ora_localize_rowptr              src/solve_ABdist.c:170-175
 * and, as the kernel-level comparator and the "port" CPU baseline, the SAME algorithm the HIP
 * path runs (same operation order, so SpMV / column-block results are bit-comparable):
 *   ora_spmv, ora_colblock_factor, ora_colblock_apply, ora_multi_dot, ora_fgmres
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORA_EXPORT __attribute__ ((visibility ("default")))

ORA_EXPORT int ora_num_threads (void)
{
#ifdef _OPENMP
   return omp_get_max_threads ();
#else
   return 1;
#endif
}

ORA_EXPORT void ora_set_num_threads (int t)
{
#ifdef _OPENMP
   if (t > 0) omp_set_num_threads (t);
#else
   (void) t;
#endif
}

/* ------------------------------------------------------------------ CSR SpMV */
/* y_r = sum over the row's entries in stored order (the HIP kernel adds in the same order) */
ORA_EXPORT void ora_spmv (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
   for (int64_t r = 0; r < n; r++) {
      double acc = 0.0;
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) acc += val[e] * x[colind[e]];
      y[r] = acc;
   }
}

ORA_EXPORT void ora_residual (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const double *x, const double *b, double *r)
{
#pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < n; i++) {
      double acc = 0.0;
      for (int32_t e = rowptr[i]; e < rowptr[i + 1]; e++) acc += val[e] * x[colind[e]];
      r[i] = b[i] - acc;
   }
}

/* componentwise backward error max_i |b-Ax|_i / (|A||x|+|b|)_i  (SuperLU's berr) */
ORA_EXPORT double ora_berr (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const double *x, const double *b)
{
   double worst = 0.0;
   for (int64_t i = 0; i < n; i++) {
      double acc = 0.0, den = 0.0;
      for (int32_t e = rowptr[i]; e < rowptr[i + 1]; e++) {
         double p = val[e] * x[colind[e]];
         acc += p;
         den += fabs (p);
      }
      den += fabs (b[i]);
      double num = fabs (b[i] - acc);
      double q = den > 0.0 ? num / den : (num > 0.0 ? 1.0e300 : 0.0);
      if (q > worst) worst = q;
   }
   return worst;
}

/* ------------------------------------------------------------------ gather / scatter / partition */
ORA_EXPORT void ora_flatten (int64_t tsl, const int32_t *ind_i, const int32_t *ind_j, const int32_t *ind_k,
                             int imt, int jmt, const double *field /* [km][jmt][imt] */, double *B /* this tracer's slice */)
{
   for (int64_t s = 0; s < tsl; s++)
      B[s] = field[((int64_t) ind_k[s] * jmt + ind_j[s]) * imt + ind_i[s]];
}

ORA_EXPORT void ora_unflatten (int64_t tsl, const int32_t *ind_i, const int32_t *ind_j, const int32_t *ind_k,
                               int imt, int jmt, const double *B, double *field)
{
   for (int64_t s = 0; s < tsl; s++)
      field[((int64_t) ind_k[s] * jmt + ind_j[s]) * imt + ind_i[s]] = B[s];
}

ORA_EXPORT void ora_rowblock_partition (int64_t n, int nprocs, int rank, int64_t *fst_row, int64_t *m_loc)
{
   int64_t base = n / nprocs;
   *fst_row = (int64_t) rank * base;
   *m_loc = (rank == nprocs - 1) ? n - *fst_row : base;
}

ORA_EXPORT void ora_localize_rowptr (int64_t m_loc, int32_t *rowptr_loc)
{
   int32_t first = rowptr_loc[0];
   for (int64_t i = 0; i <= m_loc; i++) rowptr_loc[i] -= first;
}

/* ------------------------------------------------------------------ water-column blocks */
/* fac[(d+P)*n + r]: d<0 -> l(r,r+d); d=0 -> 1/u(r,r); d>0 -> u(r,r+d).  Same elimination order
 * and the same "multiply by the reciprocal pivot" arithmetic as colblock.hip.
 * returns 0, or (row+1) of the first zero pivot */
ORA_EXPORT int64_t ora_colblock_factor (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val,
                                        int64_t nblk, const int32_t *blk_start, int P, double *fac, int *dropped)
{
   int64_t bad = 0;
   int drop = 0;
   const int W = 2 * P + 1;
#pragma omp parallel for schedule(dynamic, 64) reduction(|:drop)
   for (int64_t b = 0; b < nblk; b++) {
      const int64_t r0 = blk_start[b];
      const int len = blk_start[b + 1] - blk_start[b];
      double *a = (double *) calloc ((size_t) len * W, sizeof (double));
      for (int li = 0; li < len; li++) {
         const int64_t r = r0 + li;
         for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
            const int64_t c = colind[e];
            if (c < r0 || c >= r0 + len) continue;
            const int d = (int) (c - r);
            if (d < -P || d > P) { drop = 1; continue; }
            a[(size_t) li * W + d + P] = val[e];
         }
      }
      for (int k = 0; k < len; k++) {
         const double piv = a[(size_t) k * W + P];
         if (!(fabs (piv) > 1.0e-300)) {
#pragma omp critical
            if (bad == 0 || r0 + k + 1 < bad) bad = r0 + k + 1;
         }
         const double inv = 1.0 / piv;
         for (int dist = 1; dist <= P && k + dist < len; dist++) {
            double *row = a + (size_t) (k + dist) * W;
            const double l = row[P - dist] * inv;
            row[P - dist] = l;
            for (int q = 1; q <= P; q++) row[P - dist + q] -= l * a[(size_t) k * W + P + q];
         }
      }
      for (int li = 0; li < len; li++) {
         a[(size_t) li * W + P] = 1.0 / a[(size_t) li * W + P];
         for (int d = 0; d < W; d++) fac[(int64_t) d * n + r0 + li] = a[(size_t) li * W + d];
      }
      free (a);
   }
   if (dropped) *dropped = drop;
   return bad;
}

ORA_EXPORT void ora_colblock_apply (int64_t n, int64_t nblk, const int32_t *blk_start, int P, const double *fac, const double *rhs, double *z)
{
#pragma omp parallel for schedule(static)
   for (int64_t b = 0; b < nblk; b++) {
      const int64_t r0 = blk_start[b];
      const int len = blk_start[b + 1] - blk_start[b];
      double *y = z + r0;
      for (int li = 0; li < len; li++) y[li] = rhs[r0 + li];
      /* column-oriented forward sweep, like the kernel */
      for (int k = 0; k < len - 1; k++)
         for (int q = 1; q <= P && k + q < len; q++)
            y[k + q] -= fac[(int64_t) (P - q) * n + r0 + k + q] * y[k];
      for (int k = len - 1; k >= 0; k--) {
         y[k] *= fac[(int64_t) P * n + r0 + k];
         for (int q = 1; q <= P && k - q >= 0; q++)
            y[k - q] -= fac[(int64_t) (P + q) * n + r0 + k - q] * y[k];
      }
   }
}

/* in-block half bandwidth / longest block / rows without a diagonal */
ORA_EXPORT void ora_colblock_measure (const int32_t *rowptr, const int32_t *colind, const double *val,
                                      int64_t nblk, const int32_t *blk_start, int *out3)
{
   int bw = 0, nodiag = 0, maxlen = 0;
   for (int64_t b = 0; b < nblk; b++) {
      const int64_t r0 = blk_start[b], r1 = blk_start[b + 1];
      if (r1 - r0 > maxlen) maxlen = (int) (r1 - r0);
      for (int64_t r = r0; r < r1; r++) {
         int have = 0;
         for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
            const int64_t c = colind[e];
            if (c < r0 || c >= r1) continue;
            const int d = (int) llabs (c - r);
            if (d > bw) bw = d;
            if (c == r && val[e] != 0.0) have = 1;
         }
         nodiag += !have;
      }
   }
   out3[0] = bw;
   out3[1] = nodiag;
   out3[2] = maxlen;
}

/* ------------------------------------------------------------------ inner products */
/* out[j] = V_j . w (j<k), out[k] = w.w ; plain left-to-right sums (reference values for the
 * tolerance test of the device's tree reduction) */
ORA_EXPORT void ora_multi_dot (int64_t n, const double *V, int64_t ld, int k, const double *w, double *out)
{
   for (int j = 0; j <= k; j++) {
      const double *v = (j < k) ? V + (int64_t) j * ld : w;
      double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
      for (int64_t i = 0; i < n; i++) acc += v[i] * w[i];
      out[j] = acc;
   }
}

/* ------------------------------------------------------------------ FGMRES(m) port */
typedef struct {
   int64_t n, nblk;
   const int32_t *rowptr, *colind, *blk_start;
   const double *val;
   int P;
   double *fac;
} ora_sys;

static double dotp (int64_t n, const double *x, const double *y)
{
   double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
   for (int64_t i = 0; i < n; i++) acc += x[i] * y[i];
   return acc;
}

/* right-preconditioned flexible GMRES with two classical Gram-Schmidt passes: the algorithm of
 * csrc/solver.hip, operation for operation (reductions differ in summation order only).
 * precond: 0 none, 1 column blocks.  returns 0 converged / 1 not converged / 2 breakdown */
ORA_EXPORT int ora_fgmres (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val,
                           int64_t nblk, const int32_t *blk_start, int precond, int restart, int max_iters,
                           double rtol, const double *b, double *x, int *iters_out, double *relres_out)
{
   ora_sys S = { n, nblk, rowptr, colind, blk_start, val, 1, NULL };
   if (precond) {
      int meas[3];
      ora_colblock_measure (rowptr, colind, val, nblk, blk_start, meas);
      S.P = meas[0] <= 1 ? 1 : meas[0] <= 2 ? 2 : 4;
      S.fac = (double *) malloc ((size_t) (2 * S.P + 1) * (size_t) n * sizeof (double));
      if (!S.fac) return -2;
      if (ora_colblock_factor (n, rowptr, colind, val, nblk, blk_start, S.P, S.fac, NULL)) { free (S.fac); return -4; }
   }
   const int m = restart;
   double *V = (double *) malloc ((size_t) n * (m + 1) * sizeof (double));
   double *Z = (double *) malloc ((size_t) n * m * sizeof (double));
   double *w = (double *) malloc ((size_t) n * sizeof (double));
   double *H = (double *) calloc ((size_t) (m + 1) * m, sizeof (double));
   double *cs = (double *) malloc (sizeof (double) * m), *sn = (double *) malloc (sizeof (double) * m);
   double *g = (double *) malloc (sizeof (double) * (m + 1)), *y = (double *) malloc (sizeof (double) * m);
   double *h = (double *) malloc (sizeof (double) * (m + 2)), *h2 = (double *) malloc (sizeof (double) * (m + 2));
   int status = 1, its = 0;
   double relres = 0.0;
   const double bnorm = sqrt (dotp (n, b, b));
   memset (x, 0, (size_t) n * sizeof (double));
   if (!(bnorm > 0.0)) { status = 0; goto done; }
   const double target = rtol * bnorm;
   for (;;) {
      ora_residual (n, rowptr, colind, val, x, b, w);
      const double beta = sqrt (dotp (n, w, w));
      relres = beta / bnorm;
      if (beta <= target) { status = 0; break; }
      if (its >= max_iters) { status = 1; break; }
      const double ib = 1.0 / beta;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) V[i] = ib * w[i];
      g[0] = beta;
      int j = 0, brk = 0;
      for (; j < m && its < max_iters; j++) {
         double *vj = V + (int64_t) j * n, *zj = Z + (int64_t) j * n;
         if (precond) ora_colblock_apply (n, nblk, blk_start, S.P, S.fac, vj, zj);
         else memcpy (zj, vj, (size_t) n * sizeof (double));
         ora_spmv (n, rowptr, colind, val, zj, w);
         for (int pass = 0; pass < 2; pass++) {
            double *hh = pass ? h2 : h;
            for (int i = 0; i <= j; i++) hh[i] = dotp (n, V + (int64_t) i * n, w);
#pragma omp parallel for schedule(static)
            for (int64_t q = 0; q < n; q++) {
               double a = w[q];
               for (int i = 0; i <= j; i++) a += (-hh[i]) * V[(int64_t) i * n + q];
               w[q] = a;
            }
         }
         for (int i = 0; i <= j; i++) h[i] += h2[i];
         h[j + 1] = sqrt (dotp (n, w, w));
         const double inv = h[j + 1] > 0.0 ? 1.0 / h[j + 1] : 0.0;
         double *vn = V + (int64_t) (j + 1) * n;
#pragma omp parallel for schedule(static)
         for (int64_t q = 0; q < n; q++) vn[q] = inv * w[q];
         double *hc = H + (size_t) j * (m + 1);
         for (int i = 0; i <= j + 1; i++) hc[i] = h[i];
         for (int i = 0; i < j; i++) {
            const double t = cs[i] * hc[i] + sn[i] * hc[i + 1];
            hc[i + 1] = -sn[i] * hc[i] + cs[i] * hc[i + 1];
            hc[i] = t;
         }
         const double hjj = hc[j], hj1 = hc[j + 1], d = hypot (hjj, hj1);
         if (!(d > 0.0)) { brk = 1; break; }
         cs[j] = hjj / d;
         sn[j] = hj1 / d;
         hc[j] = d;
         hc[j + 1] = 0.0;
         g[j + 1] = -sn[j] * g[j];
         g[j] = cs[j] * g[j];
         its++;
         if (fabs (g[j + 1]) <= target || hj1 == 0.0) { j++; break; }
      }
      const int k = j;
      for (int i = k - 1; i >= 0; i--) {
         double t = g[i];
         for (int c = i + 1; c < k; c++) t -= H[(size_t) c * (m + 1) + i] * y[c];
         y[i] = t / H[(size_t) i * (m + 1) + i];
      }
#pragma omp parallel for schedule(static)
      for (int64_t q = 0; q < n; q++) {
         double a = x[q];
         for (int i = 0; i < k; i++) a += y[i] * Z[(int64_t) i * n + q];
         x[q] = a;
      }
      if (brk) {
         ora_residual (n, rowptr, colind, val, x, b, w);
         relres = sqrt (dotp (n, w, w)) / bnorm;
         status = relres * bnorm <= target ? 0 : 2;
         break;
      }
   }
 done:
   if (iters_out) *iters_out = its;
   if (relres_out) *relres_out = relres;
   free (V); free (Z); free (w); free (H); free (cs); free (sn); free (g); free (y); free (h); free (h2);
   free (S.fac);
   return status;
}

/* ------------------------------------------------------------------ direct solve (reference contract) */
/* LAPACK-style banded LU with partial pivoting on the equilibrated matrix in its natural
 * ordering, followed by double-precision iterative refinement on the ORIGINAL system until the
 * componentwise backward error stops halving (the rule of SuperLU's pdgsrfs / LAPACK dgerfs).
 * B (length n) is overwritten by X like the reference's call; returns info: 0 ok, i>0 = U(i,i)
 * exactly zero (SuperLU convention), -2 out of memory.
 * Only for sizes whose band fits memory: n * (2*kl + ku + 1) doubles. */
ORA_EXPORT int ora_direct_solve (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val,
                                 double *B, double *berr_out, int *refine_steps_out)
{
   int kl = 0, ku = 0;
   for (int64_t r = 0; r < n; r++)
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
         int64_t d = (int64_t) colind[e] - r;
         if (-d > kl) kl = (int) -d;
         if (d > ku) ku = (int) d;
      }
   /* equilibration (dgsequ-like): R_i = 1/max_j|a_ij| then C_j = 1/max_i|R_i a_ij| */
   double *R = (double *) malloc ((size_t) n * sizeof (double)), *C = (double *) calloc ((size_t) n, sizeof (double));
   if (!R || !C) return -2;
   for (int64_t r = 0; r < n; r++) {
      double mx = 0.0;
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) if (fabs (val[e]) > mx) mx = fabs (val[e]);
      R[r] = mx > 0.0 ? 1.0 / mx : 1.0;
   }
   for (int64_t r = 0; r < n; r++)
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
         double t = fabs (val[e]) * R[r];
         if (t > C[colind[e]]) C[colind[e]] = t;
      }
   for (int64_t c = 0; c < n; c++) C[c] = C[c] > 0.0 ? 1.0 / C[c] : 1.0;

   /* band storage, column-major like dgbtrf: AB[(kl+ku + i - j) + j*ldab], ldab = 2kl+ku+1 */
   const int64_t ldab = 2 * (int64_t) kl + ku + 1;
   double *AB = (double *) calloc ((size_t) (ldab * n), sizeof (double));
   int32_t *ipiv = (int32_t *) malloc ((size_t) n * sizeof (int32_t));
   double *xs = (double *) malloc ((size_t) n * sizeof (double)), *res = (double *) malloc ((size_t) n * sizeof (double));
   double *b0 = (double *) malloc ((size_t) n * sizeof (double));
   if (!AB || !ipiv || !xs || !res || !b0) return -2;
   for (int64_t r = 0; r < n; r++)
      for (int32_t e = rowptr[r]; e < rowptr[r + 1]; e++) {
         int64_t c = colind[e];
         AB[(kl + ku + r - c) + c * ldab] = R[r] * val[e] * C[c];
      }
   int info = 0;
   /* unblocked dgbtf2-style factorisation */
   for (int64_t j = 0; j < n; j++) {
      int64_t km_ = (kl < n - 1 - j) ? kl : n - 1 - j;           /* sub-diagonal rows in this column */
      double *colj = AB + j * ldab + kl + ku;                     /* diagonal element of column j */
      int64_t p = 0;
      double mx = fabs (colj[0]);
      for (int64_t i = 1; i <= km_; i++) if (fabs (colj[i]) > mx) { mx = fabs (colj[i]); p = i; }
      ipiv[j] = (int32_t) (j + p);
      if (mx == 0.0) { if (!info) info = (int) (j + 1); continue; }
      if (p) for (int64_t c = j; c <= (j + ku + kl < n - 1 ? j + ku + kl : n - 1); c++) {
            double *a1 = AB + c * ldab + (kl + ku + j - c), *a2 = AB + c * ldab + (kl + ku + j + p - c);
            double t = *a1; *a1 = *a2; *a2 = t;
         }
      const double inv = 1.0 / colj[0];
      for (int64_t i = 1; i <= km_; i++) colj[i] *= inv;
      int64_t clast = j + ku + kl < n - 1 ? j + ku + kl : n - 1;
      for (int64_t c = j + 1; c <= clast; c++) {
         double *cc = AB + c * ldab + (kl + ku + j - c);            /* element (j, c) */
         const double ujc = cc[0];
         if (ujc != 0.0) for (int64_t i = 1; i <= km_; i++) cc[i] -= colj[i] * ujc;
      }
   }
   memcpy (b0, B, (size_t) n * sizeof (double));
   memset (xs, 0, (size_t) n * sizeof (double));
   double berr = 1.0e300, last = 1.0e300;
   int steps = 0;
   if (!info) for (int it = 0; it < 20; it++) {
         /* residual of the ORIGINAL system and its componentwise backward error */
         ora_residual (n, rowptr, colind, val, xs, b0, res);
         berr = (it == 0) ? 1.0e300 : ora_berr (n, rowptr, colind, val, xs, b0);
         if (it > 0 && (berr <= 2.2204460492503131e-16 || berr > 0.5 * last)) break;
         last = berr;
         /* dx = C * (LU)^-1 * (R * res) */
         for (int64_t i = 0; i < n; i++) res[i] *= R[i];
         for (int64_t j = 0; j < n; j++) {                           /* L solve with row interchanges */
            int64_t p = ipiv[j];
            if (p != j) { double t = res[j]; res[j] = res[p]; res[p] = t; }
            int64_t km_ = (kl < n - 1 - j) ? kl : n - 1 - j;
            const double *colj = AB + j * ldab + kl + ku;
            const double rj = res[j];
            if (rj != 0.0) for (int64_t i = 1; i <= km_; i++) res[j + i] -= colj[i] * rj;
         }
         for (int64_t j = n - 1; j >= 0; j--) {                      /* U solve (upper bandwidth kl+ku) */
            const double *colj = AB + j * ldab + kl + ku;
            res[j] /= colj[0];
            const double xj = res[j];
            int64_t top = j - (kl + ku) > 0 ? j - (kl + ku) : 0;
            if (xj != 0.0) for (int64_t i = top; i < j; i++) res[i] -= AB[j * ldab + (kl + ku + i - j)] * xj;
         }
         for (int64_t i = 0; i < n; i++) xs[i] += C[i] * res[i];
         steps = it;
      }
   if (!info) {
      berr = ora_berr (n, rowptr, colind, val, xs, b0);
      memcpy (B, xs, (size_t) n * sizeof (double));
   }
   if (berr_out) *berr_out = berr;
   if (refine_steps_out) *refine_steps_out = steps;
   free (R); free (C); free (AB); free (ipiv); free (xs); free (res); free (b0);
   return info;
}
