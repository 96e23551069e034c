/* ml_oracle.c -- CPU ORACLE of the multilevel water-column preconditioner and the FGMRES iteration around it.
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product never links, imports or calls it and has no CPU fallback.
 *
 * PARITY UNPINNED by the reference (see nkp_oracle.c: the reference holds no vectors and its arithmetic -- SuperLU_DIST
 * 5.1.3, reference src/Makefile:3 -- is not in the tree).  What this file restates is therefore not a reference routine but
 * the ALGORITHM the HIP path substitutes for the reference's factor + solve calls (pdgssvx_ABglobal with nrhs = 0 and
 * nrhs = 1, reference src/solve_ABglobal.c:349-360 and :393-402), from its description in DESIGN.md section 2:
 *   - low-order twin L of A: D_ij = max (0, -a_ij, -a_ji) between water columns, L = A + D - diag (rowsum D)
 *   - hierarchy: columns grouped 2 x 2 in (i, j) (4 x 4 from level 3 down on grids below 200 000 columns), inside a group
 *     one coarse cell per laterally connected set of same-depth cells, same-depth sets of <= 4 cells merged across groups,
 *     sets threaded through depth into coarse columns by largest overlap, leaf stubs absorbed; piecewise-constant P,
 *     Galerkin operators P^T L P; dense inverse on the last level (<= 3000 rows)
 *   - smoother: 2-colour ((i + j) & 1) water-column block Gauss-Seidel, 3 sweeps before (colour 0 first) and 3 after
 *     (colour 1 first) the coarse correction, which is weighted by 1.1
 *   - right-preconditioned flexible GMRES (200), one classical Gram-Schmidt pass, true residual at every restart
 * It is pinned by tests/test_oracle.py against the independent scipy restatement tests/ml_reference.py (same coarse cells,
 * same cycle to rounding) and is the `"kind": "port"` CPU baseline of bench.py: the same algorithm on all host cores.
 * Water columns are contiguous runs of rows, k innermost (reference src/matrix.c:239-251).
 *
 * Build: oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off), same library as nkp_oracle.c.
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

/* from nkp_oracle.c */
int64_t ora_colblock_factor (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, int64_t nblk, const int32_t *blk_start, int P, double *fac, int *dropped);
void ora_colblock_measure (const int32_t *rowptr, const int32_t *colind, const double *val, int64_t nblk, const int32_t *blk_start, int *out3);

typedef struct {
   int64_t n;
   int32_t *rowptr, *colind;
   double *val;
} csr_t;

typedef struct {
   csr_t A;                        /* level operator, natural (water-column) order */
   int ncol;
   int32_t *blk_start, *ktop, *ci, *cj, *ct;   /* per column: first row, depth of the first row, grid position, tracer */
   int32_t *col_of;                /* per row */
   int P;
   double *fac;                    /* band LU of the column blocks, fac[(d + P) * n + row] */
   int32_t *cmap;                  /* row -> row of the next level */
   int32_t *rptr, *ridx;           /* row of the next level -> its rows here (ascending) */
   int64_t nc;
   double *x, *b, *r;
   double *dense_inv;              /* last level */
} level_t;

typedef struct ora_ml {
   int nlev, nu;
   double omega;
   level_t *lev;
} ora_ml;

static void *xmalloc (size_t b) { void *p = malloc (b ? b : 1); if (!p) { fprintf (stderr, "ml_oracle: out of memory\n"); abort (); } return p; }
static void *xcalloc (size_t k, size_t b) { void *p = calloc (k ? k : 1, b); if (!p) { fprintf (stderr, "ml_oracle: out of memory\n"); abort (); } return p; }

static void csr_free (csr_t *A) { free (A->rowptr); free (A->colind); free (A->val); memset (A, 0, sizeof *A); }

/* ------------------------------------------------------------------ low-order twin */
static double entry_of (const csr_t *A, int64_t r, int32_t c)
{
   int32_t lo = A->rowptr[r], hi = A->rowptr[r + 1];
   while (lo < hi) {
      const int32_t mid = lo + (hi - lo) / 2;
      if (A->colind[mid] < c) lo = mid + 1;
      else hi = mid;
   }
   return (lo < A->rowptr[r + 1] && A->colind[lo] == c) ? A->val[lo] : 0.0;
}

static csr_t low_order_twin (const csr_t *A, const int32_t *col_of)
{
   const int64_t n = A->n;
   double *nv = xmalloc ((size_t) A->rowptr[n] * sizeof (double));
   int32_t *keep = xcalloc ((size_t) n + 1, sizeof (int32_t));
#pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < n; i++) {
      double dsum = 0.0;
      int32_t dpos = -1, cnt = 0;
      for (int32_t e = A->rowptr[i]; e < A->rowptr[i + 1]; e++) {
         const int32_t j = A->colind[e];
         double a = A->val[e];
         if (j == i) { dpos = e; nv[e] = a; cnt++; continue; }
         if (col_of[j] != col_of[i]) {
            const double aji = entry_of (A, j, (int32_t) i);
            double d = 0.0;
            if (-a > d) d = -a;
            if (-aji > d) d = -aji;
            a += d;
            dsum += d;
         }
         nv[e] = a;
         if (a != 0.0 || col_of[j] == col_of[i]) cnt++;
      }
      if (dpos >= 0) nv[dpos] -= dsum;
      keep[i + 1] = cnt;
   }
   csr_t L;
   L.n = n;
   L.rowptr = keep;
   for (int64_t i = 0; i < n; i++) L.rowptr[i + 1] += L.rowptr[i];
   L.colind = xmalloc ((size_t) L.rowptr[n] * sizeof (int32_t));
   L.val = xmalloc ((size_t) L.rowptr[n] * sizeof (double));
#pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < n; i++) {
      int32_t q = L.rowptr[i];
      for (int32_t e = A->rowptr[i]; e < A->rowptr[i + 1]; e++) {
         const int32_t j = A->colind[e];
         if (j != i && col_of[j] != col_of[i] && nv[e] == 0.0) continue;      /* a removed coupling is not stored */
         L.colind[q] = j;
         L.val[q] = nv[e];
         q++;
      }
   }
   free (nv);
   return L;
}

/* ------------------------------------------------------------------ coarse cells of one coarsening step */
static int32_t uf_find (int32_t *p, int32_t x)
{
   while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
   return x;
}
static void uf_unite (int32_t *p, int32_t a, int32_t b)
{
   a = uf_find (p, a);
   b = uf_find (p, b);
   if (a == b) return;
   if (a < b) p[b] = a;                      /* the lowest row of a set is its root */
   else p[a] = b;
}

typedef struct { int64_t key; int32_t col; } keyed_t;
static int keyed_cmp (const void *a, const void *b)
{
   const keyed_t *x = a, *y = b;
   if (x->key != y->key) return x->key < y->key ? -1 : 1;
   return x->col < y->col ? -1 : x->col > y->col;
}
static int int_cmp (const void *a, const void *b) { const int32_t x = *(const int32_t *) a, y = *(const int32_t *) b; return x < y ? -1 : x > y; }

/* F = fine level (operator, columns); sh = 1 (2 x 2 groups) or 2 (4 x 4); fills F->cmap / F->nc and the column arrays of C */
static void coarse_cells (level_t *F, level_t *C, int sh, int pocket, double tau)
{
   const csr_t *L = &F->A;
   const int64_t n = L->n;
   const int ncol = F->ncol;
   const int32_t *col_of = F->col_of, *blk = F->blk_start, *ktop = F->ktop;
#define DEPTH(r) (ktop[col_of[r]] + ((int32_t) (r) - blk[col_of[r]]))
#define ROW_AT(c, k) (((k) >= ktop[c] && blk[c] + ((k) - ktop[c]) < blk[(c) + 1]) ? blk[c] + ((k) - ktop[c]) : -1)
   /* groups of columns: same tracer, same (j >> sh, i >> sh) */
   int32_t *group = xmalloc ((size_t) ncol * sizeof (int32_t));
   {
      keyed_t *ks = xmalloc ((size_t) ncol * sizeof (keyed_t));
      for (int c = 0; c < ncol; c++) {
         ks[c].key = (((int64_t) F->ct[c] << 40) | ((int64_t) (F->cj[c] >> sh) << 20)) | (int64_t) (F->ci[c] >> sh);
         ks[c].col = c;
      }
      qsort (ks, (size_t) ncol, sizeof (keyed_t), keyed_cmp);
      int32_t g = -1;
      for (int q = 0; q < ncol; q++) {
         if (q == 0 || ks[q].key != ks[q - 1].key) g++;
         group[ks[q].col] = g;
      }
      free (ks);
   }
   /* leaf stubs: columns that start below the surface and that no outside row feels */
   char *leaf = xcalloc ((size_t) ncol, 1);
   int32_t *anchor = xmalloc ((size_t) ncol * sizeof (int32_t));
   for (int c = 0; c < ncol; c++) anchor[c] = -1;
   int have_stubs = 0;
   for (int c = 0; c < ncol; c++) have_stubs |= ktop[c] > 0;
   if (have_stubs && tau > 0.0) {
      double *felt = xcalloc ((size_t) ncol, sizeof (double)), *best = xmalloc ((size_t) ncol * sizeof (double));
      for (int c = 0; c < ncol; c++) best[c] = -1.0;
      for (int64_t r = 0; r < n; r++) {
         const int32_t c = col_of[r];
         const double dg = fabs (entry_of (L, r, (int32_t) r));
         for (int32_t e = L->rowptr[r]; e < L->rowptr[r + 1]; e++) {
            const int32_t j = L->colind[e], c2 = col_of[j];
            if (c2 == c || F->ct[c2] != F->ct[c]) continue;
            if (ktop[c2] == 0 && ktop[c] == 0) continue;
            const double v = fabs (L->val[e]), f = dg > 0.0 ? v / dg : 1.0e300;
            if (f > felt[c2]) felt[c2] = f;
            if (v >= best[c]) { best[c] = v; anchor[c] = j; }          /* strongest coupling of the column, ties -> later entry */
         }
      }
      for (int c = 0; c < ncol; c++) leaf[c] = ktop[c] > 0 && felt[c] < tau && anchor[c] >= 0;
      char *both = xcalloc ((size_t) ncol, 1);
      for (int c = 0; c < ncol; c++) both[c] = leaf[c] && leaf[col_of[anchor[c]]];
      for (int c = 0; c < ncol; c++) if (both[c]) leaf[c] = 0;
      free (both); free (felt); free (best);
   }
   /* lateral edges between cells of equal depth: all of them find the pockets, those inside a group (and those of
    * the pockets) make the coarse cells */
   int32_t *U = xmalloc ((size_t) n * sizeof (int32_t)), *U0 = xmalloc ((size_t) n * sizeof (int32_t));
   for (int64_t r = 0; r < n; r++) U[r] = U0[r] = (int32_t) r;
   for (int pass = 0; pass < 2; pass++) {
      int32_t *size0 = NULL;
      if (pass == 1) {
         if (pocket <= 0) break;
         size0 = xcalloc ((size_t) n, sizeof (int32_t));
         for (int64_t r = 0; r < n; r++) size0[uf_find (U0, (int32_t) r)]++;
      }
      for (int64_t r = 0; r < n; r++) {
         const int32_t c = col_of[r];
         if (leaf[c]) continue;
         if (pass == 1) { const int32_t s = size0[uf_find (U0, (int32_t) r)]; if (s > pocket || s <= 1) continue; }
         const int32_t k = DEPTH (r);
         for (int32_t e = L->rowptr[r]; e < L->rowptr[r + 1]; e++) {
            const int32_t j = L->colind[e], c2 = col_of[j];
            if (c2 == c || leaf[c2] || F->ct[c2] != F->ct[c]) continue;
            const int32_t dk = DEPTH (j) - k;
            if (dk < -1 || dk > 1) continue;
            const int32_t t = ROW_AT (c2, k);
            if (t < 0) continue;
            if (pass == 0) {
               uf_unite (U0, (int32_t) r, t);
               if (group[c] == group[c2]) uf_unite (U, (int32_t) r, t);
            } else if (group[c] != group[c2])
               uf_unite (U, (int32_t) r, t);
         }
      }
      free (size0);
   }
   /* sets numbered by their lowest row */
   int32_t *comp = xmalloc ((size_t) n * sizeof (int32_t));
   int32_t ncomp = 0;
   for (int64_t r = 0; r < n; r++) {
      const int32_t root = uf_find (U, (int32_t) r);
      if (root == r) comp[r] = ncomp++;
      else comp[r] = comp[root];
   }
   int32_t *kcomp = xmalloc ((size_t) ncomp * sizeof (int32_t));
   for (int64_t r = 0; r < n; r++) kcomp[comp[r]] = DEPTH (r);
   /* overlaps between a set and the sets directly below it */
   int32_t *bptr = xcalloc ((size_t) ncomp + 1, sizeof (int32_t));
   for (int c = 0; c < ncol; c++)
      for (int32_t r = blk[c]; r + 1 < blk[c + 1]; r++) bptr[comp[r] + 1]++;
   for (int32_t q = 0; q < ncomp; q++) bptr[q + 1] += bptr[q];
   int32_t *child = xmalloc ((size_t) bptr[ncomp] * sizeof (int32_t)), *fill = xmalloc ((size_t) (ncomp + 1) * sizeof (int32_t));
   memcpy (fill, bptr, (size_t) (ncomp + 1) * sizeof (int32_t));
   for (int c = 0; c < ncol; c++)
      for (int32_t r = blk[c]; r + 1 < blk[c + 1]; r++) child[fill[comp[r]]++] = comp[r + 1];
   int32_t *bestpar = xmalloc ((size_t) ncomp * sizeof (int32_t)), *bestpar_cnt = xcalloc ((size_t) ncomp, sizeof (int32_t));
   int32_t *bestchi = xmalloc ((size_t) ncomp * sizeof (int32_t));
   for (int32_t q = 0; q < ncomp; q++) bestpar[q] = bestchi[q] = -1;
   for (int32_t par = 0; par < ncomp; par++) {
      int32_t *b0 = child + bptr[par], *b1 = child + bptr[par + 1];
      qsort (b0, (size_t) (b1 - b0), sizeof (int32_t), int_cmp);
      int32_t bc = 0;
      for (int32_t *q = b0; q < b1;) {
         int32_t *q2 = q;
         while (q2 < b1 && *q2 == *q) q2++;
         const int32_t chi = *q, cnt = (int32_t) (q2 - q);
         if (cnt > bestpar_cnt[chi]) { bestpar_cnt[chi] = cnt; bestpar[chi] = par; }    /* ties: the lower parent */
         if (cnt > bc) { bc = cnt; bestchi[par] = chi; }                                 /* ties: the lower child */
         q = q2;
      }
   }
   /* threading: sets in order of (depth, id); the child with the largest overlap continues its parent's column */
   int32_t kmax = 0;
   for (int32_t q = 0; q < ncomp; q++) if (kcomp[q] > kmax) kmax = kcomp[q];
   int32_t *kptr = xcalloc ((size_t) kmax + 2, sizeof (int32_t)), *order = xmalloc ((size_t) ncomp * sizeof (int32_t));
   for (int32_t q = 0; q < ncomp; q++) kptr[kcomp[q] + 1]++;
   for (int32_t k = 0; k <= kmax; k++) kptr[k + 1] += kptr[k];
   for (int32_t q = 0; q < ncomp; q++) order[kptr[kcomp[q]]++] = q;
   int32_t *ccol = xmalloc ((size_t) ncomp * sizeof (int32_t)), *cc_ktop = xmalloc ((size_t) ncomp * sizeof (int32_t)), *cc_len = xcalloc ((size_t) ncomp, sizeof (int32_t));
   int32_t nraw = 0;
   for (int32_t o = 0; o < ncomp; o++) {
      const int32_t id = order[o], par = bestpar[id];
      if (par >= 0 && bestchi[par] == id) ccol[id] = ccol[par];
      else { ccol[id] = nraw; cc_ktop[nraw] = kcomp[id]; nraw++; }
      cc_len[ccol[id]]++;
   }
   /* a coarse column sits at the position of the group of its lowest fine row; absorbed stubs own no coarse column */
   int32_t *first_row = xmalloc ((size_t) (nraw ? nraw : 1) * sizeof (int32_t));
   for (int32_t q = 0; q < nraw; q++) first_row[q] = -1;
   for (int64_t r = 0; r < n; r++)
      if (!leaf[col_of[r]] && first_row[ccol[comp[r]]] < 0) first_row[ccol[comp[r]]] = (int32_t) r;
   int32_t *newid = xmalloc ((size_t) (nraw ? nraw : 1) * sizeof (int32_t));
   int ncc = 0;
   for (int32_t q = 0; q < nraw; q++) newid[q] = first_row[q] >= 0 ? ncc++ : -1;
   C->ncol = ncc;
   C->blk_start = xcalloc ((size_t) ncc + 1, sizeof (int32_t));
   C->ktop = xmalloc ((size_t) ncc * sizeof (int32_t));
   C->ci = xmalloc ((size_t) ncc * sizeof (int32_t));
   C->cj = xmalloc ((size_t) ncc * sizeof (int32_t));
   C->ct = xmalloc ((size_t) ncc * sizeof (int32_t));
   for (int32_t q = 0; q < nraw; q++) {
      const int32_t a = newid[q];
      if (a < 0) continue;
      const int32_t c = col_of[first_row[q]];
      C->blk_start[a + 1] = cc_len[q];
      C->ktop[a] = cc_ktop[q];
      C->ci[a] = F->ci[c] >> sh;
      C->cj[a] = F->cj[c] >> sh;
      C->ct[a] = F->ct[c];
   }
   for (int a = 0; a < ncc; a++) C->blk_start[a + 1] += C->blk_start[a];
   F->nc = C->blk_start[ncc];
   F->cmap = xmalloc ((size_t) n * sizeof (int32_t));
   for (int64_t r = 0; r < n; r++) {
      if (leaf[col_of[r]]) continue;
      const int32_t a = newid[ccol[comp[r]]];
      F->cmap[r] = C->blk_start[a] + (DEPTH (r) - C->ktop[a]);
   }
   for (int c = 0; c < ncol; c++)
      if (leaf[c])
         for (int32_t r = blk[c]; r < blk[c + 1]; r++) F->cmap[r] = F->cmap[anchor[c]];
   C->col_of = xmalloc ((size_t) F->nc * sizeof (int32_t));
   for (int a = 0; a < ncc; a++)
      for (int32_t r = C->blk_start[a]; r < C->blk_start[a + 1]; r++) C->col_of[r] = a;
   free (group); free (leaf); free (anchor); free (U); free (U0); free (comp); free (kcomp); free (bptr); free (child); free (fill);
   free (bestpar); free (bestpar_cnt); free (bestchi); free (kptr); free (order); free (ccol); free (cc_ktop); free (cc_len); free (first_row); free (newid);
#undef DEPTH
#undef ROW_AT
}

/* ------------------------------------------------------------------ Galerkin product P^T L P, P piecewise constant */
static csr_t galerkin (const csr_t *L, const int32_t *cmap, int64_t nc)
{
   int32_t *rptr = xcalloc ((size_t) nc + 1, sizeof (int32_t)), *ridx = xmalloc ((size_t) L->n * sizeof (int32_t));
   for (int64_t i = 0; i < L->n; i++) rptr[cmap[i] + 1]++;
   for (int64_t I = 0; I < nc; I++) rptr[I + 1] += rptr[I];
   int32_t *fill = xmalloc ((size_t) (nc + 1) * sizeof (int32_t));
   memcpy (fill, rptr, (size_t) (nc + 1) * sizeof (int32_t));
   for (int64_t i = 0; i < L->n; i++) ridx[fill[cmap[i]]++] = (int32_t) i;
   free (fill);
   /* two passes over the coarse rows (count, fill), each thread with its own accumulator over the coarse columns */
   csr_t C;
   C.n = nc;
   C.rowptr = xcalloc ((size_t) nc + 1, sizeof (int32_t));
   C.colind = NULL;
   C.val = NULL;
   for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) {
         for (int64_t I = 0; I < nc; I++) C.rowptr[I + 1] += C.rowptr[I];
         C.colind = xmalloc ((size_t) C.rowptr[nc] * sizeof (int32_t));
         C.val = xmalloc ((size_t) C.rowptr[nc] * sizeof (double));
      }
#pragma omp parallel
      {
         double *acc = xcalloc ((size_t) nc, sizeof (double));
         char *mark = xcalloc ((size_t) nc, 1);
         int32_t *touched = xmalloc ((size_t) 65536 * sizeof (int32_t));
         size_t cap = 65536;
#pragma omp for schedule(dynamic, 256)
         for (int64_t I = 0; I < nc; I++) {
            size_t nt = 0;
            for (int32_t q = rptr[I]; q < rptr[I + 1]; q++) {
               const int32_t i = ridx[q];
               for (int32_t e = L->rowptr[i]; e < L->rowptr[i + 1]; e++) {
                  const int32_t J = cmap[L->colind[e]];
                  if (!mark[J]) {
                     mark[J] = 1;
                     if (nt == cap) { cap *= 2; touched = realloc (touched, cap * sizeof (int32_t)); if (!touched) abort (); }
                     touched[nt++] = J;
                  }
                  acc[J] += L->val[e];
               }
            }
            qsort (touched, nt, sizeof (int32_t), int_cmp);
            int32_t cnt = 0, o = pass ? C.rowptr[I] : 0;
            for (size_t t = 0; t < nt; t++) {
               const int32_t J = touched[t];
               if (acc[J] != 0.0 || J == I) {
                  if (pass) { C.colind[o + cnt] = J; C.val[o + cnt] = acc[J]; }
                  cnt++;
               }
               acc[J] = 0.0;
               mark[J] = 0;
            }
            if (!pass) C.rowptr[I + 1] = cnt;
         }
         free (acc); free (mark); free (touched);
      }
   }
   free (rptr); free (ridx);
   return C;
}

/* dense inverse by Gauss-Jordan with partial pivoting; returns 0, or 1 if singular */
static int dense_inverse (int n, double *a, double *inv)
{
   memset (inv, 0, (size_t) n * n * sizeof (double));
   for (int i = 0; i < n; i++) inv[(size_t) i * n + i] = 1.0;
   for (int k = 0; k < n; k++) {
      int p = k;
      double mx = fabs (a[(size_t) k * n + k]);
      for (int i = k + 1; i < n; i++) if (fabs (a[(size_t) i * n + k]) > mx) { mx = fabs (a[(size_t) i * n + k]); p = i; }
      if (!(mx > 0.0)) return 1;
      if (p != k)
         for (int c = 0; c < n; c++) {
            double t = a[(size_t) k * n + c]; a[(size_t) k * n + c] = a[(size_t) p * n + c]; a[(size_t) p * n + c] = t;
            t = inv[(size_t) k * n + c]; inv[(size_t) k * n + c] = inv[(size_t) p * n + c]; inv[(size_t) p * n + c] = t;
         }
      const double piv = 1.0 / a[(size_t) k * n + k];
      for (int c = 0; c < n; c++) { a[(size_t) k * n + c] *= piv; inv[(size_t) k * n + c] *= piv; }
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++) {
         if (i == k) continue;
         const double f = a[(size_t) i * n + k];
         if (f == 0.0) continue;
         double *ai = a + (size_t) i * n, *ak = a + (size_t) k * n, *ii = inv + (size_t) i * n, *ik = inv + (size_t) k * n;
         for (int c = 0; c < n; c++) { ai[c] -= f * ak[c]; ii[c] -= f * ik[c]; }
      }
   }
   return 0;
}

/* ------------------------------------------------------------------ setup */
ORA_EXPORT void ora_ml_free (ora_ml *M)
{
   if (!M) return;
   for (int l = 0; l < M->nlev; l++) {
      level_t *V = &M->lev[l];
      csr_free (&V->A);
      free (V->blk_start); free (V->ktop); free (V->ci); free (V->cj); free (V->ct); free (V->col_of); free (V->fac); free (V->cmap);
      free (V->x); free (V->b); free (V->r); free (V->dense_inv); free (V->rptr); free (V->ridx);
   }
   free (M->lev);
   free (M);
}

/* col_i / col_j: grid position of every water column (nblk entries); returns NULL with a message on stderr on failure */
ORA_EXPORT ora_ml *ora_ml_setup (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, int64_t nblk, const int32_t *blk_start,
                                 const int32_t *col_i, const int32_t *col_j, int coupled_tracer_cnt, int nu, int coarsest_rows, int max_levels)
{
   ora_ml *M = xcalloc (1, sizeof (ora_ml));
   M->nu = nu > 0 ? nu : 3;
   M->omega = 1.1;
   if (max_levels <= 0) max_levels = 12;
   M->lev = xcalloc ((size_t) max_levels, sizeof (level_t));
   level_t *V = &M->lev[0];
   V->ncol = (int) nblk;
   V->blk_start = xmalloc ((size_t) (nblk + 1) * sizeof (int32_t));
   memcpy (V->blk_start, blk_start, (size_t) (nblk + 1) * sizeof (int32_t));
   V->ktop = xcalloc ((size_t) nblk, sizeof (int32_t));
   V->ci = xmalloc ((size_t) nblk * sizeof (int32_t));
   V->cj = xmalloc ((size_t) nblk * sizeof (int32_t));
   V->ct = xmalloc ((size_t) nblk * sizeof (int32_t));
   const int64_t per = (coupled_tracer_cnt > 1 && nblk % coupled_tracer_cnt == 0) ? nblk / coupled_tracer_cnt : nblk;
   for (int64_t c = 0; c < nblk; c++) { V->ci[c] = col_i[c]; V->cj[c] = col_j[c]; V->ct[c] = (int32_t) (c / per); }   /* tracer-major rows: reference src/matrix.c:778-784 */
   V->col_of = xmalloc ((size_t) n * sizeof (int32_t));
   for (int64_t c = 0; c < nblk; c++)
      for (int32_t r = blk_start[c]; r < blk_start[c + 1]; r++) V->col_of[r] = (int32_t) c;
   {
      csr_t A = { n, (int32_t *) rowptr, (int32_t *) colind, (double *) val };
      V->A = low_order_twin (&A, V->col_of);
   }
   const int big_from = (nblk / (coupled_tracer_cnt > 0 ? coupled_tracer_cnt : 1) >= 200000) ? -1 : 3;
   int l = 0;
   for (;; l++) {
      V = &M->lev[l];
      const int last = (l + 1 >= max_levels) || V->A.n <= coarsest_rows || V->ncol <= 4;
      if (last) break;
      level_t *C = &M->lev[l + 1];
      coarse_cells (V, C, (big_from >= 0 && l >= big_from) ? 2 : 1, 4, 0.01);
      if (V->nc >= V->A.n) {                           /* no coarsening possible */
         free (C->blk_start); free (C->ktop); free (C->ci); free (C->cj); free (C->ct); free (C->col_of);
         memset (C, 0, sizeof *C);
         free (V->cmap);
         V->cmap = NULL;
         break;
      }
      C->A = galerkin (&V->A, V->cmap, V->nc);
   }
   M->nlev = l + 1;
   for (l = 0; l < M->nlev; l++) {
      V = &M->lev[l];
      const int64_t nl = V->A.n;
      V->x = xcalloc ((size_t) nl, sizeof (double));
      V->b = xcalloc ((size_t) nl, sizeof (double));
      V->r = xcalloc ((size_t) nl, sizeof (double));
      if (l == M->nlev - 1) {
         double *a = xcalloc ((size_t) nl * nl, sizeof (double));
         for (int64_t i = 0; i < nl; i++)
            for (int32_t e = V->A.rowptr[i]; e < V->A.rowptr[i + 1]; e++) a[(size_t) i * nl + V->A.colind[e]] = V->A.val[e];
         V->dense_inv = xmalloc ((size_t) nl * nl * sizeof (double));
         const int bad = dense_inverse ((int) nl, a, V->dense_inv);
         free (a);
         if (bad) { fprintf (stderr, "ml_oracle: coarsest operator is singular\n"); ora_ml_free (M); return NULL; }
      } else {
         V->rptr = xcalloc ((size_t) V->nc + 1, sizeof (int32_t));
         V->ridx = xmalloc ((size_t) nl * sizeof (int32_t));
         for (int64_t i = 0; i < nl; i++) V->rptr[V->cmap[i] + 1]++;
         for (int64_t I = 0; I < V->nc; I++) V->rptr[I + 1] += V->rptr[I];
         {
            int32_t *fill = xmalloc ((size_t) (V->nc + 1) * sizeof (int32_t));
            memcpy (fill, V->rptr, (size_t) (V->nc + 1) * sizeof (int32_t));
            for (int64_t i = 0; i < nl; i++) V->ridx[fill[V->cmap[i]]++] = (int32_t) i;
            free (fill);
         }
         int meas[3];
         ora_colblock_measure (V->A.rowptr, V->A.colind, V->A.val, V->ncol, V->blk_start, meas);
         if (meas[1]) { fprintf (stderr, "ml_oracle: level %d has rows without a diagonal\n", l); ora_ml_free (M); return NULL; }
         V->P = meas[0] <= 1 ? 1 : meas[0] <= 2 ? 2 : 4;
         V->fac = xmalloc ((size_t) (2 * V->P + 1) * (size_t) nl * sizeof (double));
         if (ora_colblock_factor (nl, V->A.rowptr, V->A.colind, V->A.val, V->ncol, V->blk_start, V->P, V->fac, NULL)) {
            fprintf (stderr, "ml_oracle: zero pivot in a column block of level %d\n", l);
            ora_ml_free (M);
            return NULL;
         }
      }
   }
   return M;
}

ORA_EXPORT int ora_ml_levels (const ora_ml *M) { return M->nlev; }
ORA_EXPORT int64_t ora_ml_level_rows (const ora_ml *M, int l) { return M->lev[l].A.n; }
ORA_EXPORT int64_t ora_ml_level_nnz (const ora_ml *M, int l) { return M->lev[l].A.rowptr[M->lev[l].A.n]; }
/* coarse row of every row of level l (l < levels - 1) and column of every row of level l */
ORA_EXPORT void ora_ml_level_maps (const ora_ml *M, int l, int32_t *cmap, int32_t *col_of)
{
   const level_t *V = &M->lev[l];
   if (cmap && V->cmap) memcpy (cmap, V->cmap, (size_t) V->A.n * sizeof (int32_t));
   if (col_of) memcpy (col_of, V->col_of, (size_t) V->A.n * sizeof (int32_t));
}

/* ------------------------------------------------------------------ cycle */
/* one half sweep over the columns of one colour: residual of all their rows first (columns of one colour are coupled through
 * the +-2 neighbours of upwind3 and through stub columns), then the band solves, x += z */
static void half_sweep (level_t *V, int colour, int first)
{
   const csr_t *A = &V->A;
   const int P = V->P;
   const int64_t n = A->n;
   if (!first) {
#pragma omp parallel for schedule(dynamic, 64)
      for (int c = 0; c < V->ncol; c++) {
         if (((V->ci[c] + V->cj[c]) & 1) != colour) continue;
         for (int32_t r = V->blk_start[c]; r < V->blk_start[c + 1]; r++) {
            double acc = 0.0;
            for (int32_t e = A->rowptr[r]; e < A->rowptr[r + 1]; e++) acc += A->val[e] * V->x[A->colind[e]];
            V->r[r] = V->b[r] - acc;
         }
      }
   }
   const double *rhs = first ? V->b : V->r;            /* x = 0: the residual is b */
#pragma omp parallel for schedule(dynamic, 64)
   for (int c = 0; c < V->ncol; c++) {
      if (((V->ci[c] + V->cj[c]) & 1) != colour) continue;
      const int64_t r0 = V->blk_start[c];
      const int len = V->blk_start[c + 1] - V->blk_start[c];
      double y[256];
      for (int k = 0; k < len; k++) y[k] = rhs[r0 + k];
      for (int k = 0; k < len; k++) {                   /* forward, far diagonal first */
         double t = y[k];
         for (int q = P; q >= 1; q--) if (k - q >= 0) t -= V->fac[(int64_t) (P - q) * n + r0 + k] * y[k - q];
         y[k] = t;
      }
      for (int k = len - 1; k >= 0; k--) {
         double t = y[k];
         for (int q = P; q >= 1; q--) if (k + q < len) t -= V->fac[(int64_t) (P + q) * n + r0 + k] * y[k + q];
         y[k] = t * V->fac[(int64_t) P * n + r0 + k];
      }
      for (int k = 0; k < len; k++) V->x[r0 + k] += y[k];
   }
}

static void cycle (ora_ml *M, int l)
{
   level_t *V = &M->lev[l];
   const int64_t n = V->A.n;
   if (l == M->nlev - 1) {
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) {
         double acc = 0.0;
         const double *row = V->dense_inv + (size_t) i * n;
         for (int64_t j = 0; j < n; j++) acc += row[j] * V->b[j];
         V->x[i] = acc;
      }
      return;
   }
   memset (V->x, 0, (size_t) n * sizeof (double));
   half_sweep (V, 0, 1);
   half_sweep (V, 1, 0);
   for (int s = 1; s < M->nu; s++) { half_sweep (V, 0, 0); half_sweep (V, 1, 0); }
   level_t *C = &M->lev[l + 1];
   const csr_t *A = &V->A;
#pragma omp parallel for schedule(static)
   for (int64_t r = 0; r < n; r++) {
      double acc = 0.0;
      for (int32_t e = A->rowptr[r]; e < A->rowptr[r + 1]; e++) acc += A->val[e] * V->x[A->colind[e]];
      V->r[r] = V->b[r] - acc;
   }
#pragma omp parallel for schedule(static)
   for (int64_t I = 0; I < V->nc; I++) {                                  /* restriction = P^T, ascending fine rows */
      double acc = 0.0;
      for (int32_t q = V->rptr[I]; q < V->rptr[I + 1]; q++) acc += V->r[V->ridx[q]];
      C->b[I] = acc;
   }
   cycle (M, l + 1);
#pragma omp parallel for schedule(static)
   for (int64_t r = 0; r < n; r++) V->x[r] += M->omega * C->x[V->cmap[r]];
   for (int s = 0; s < M->nu; s++) { half_sweep (V, 1, 0); half_sweep (V, 0, 0); }
}

/* z = V(nu, nu)-cycle (r) */
ORA_EXPORT void ora_ml_apply (ora_ml *M, const double *r, double *z)
{
   level_t *V = &M->lev[0];
   memcpy (V->b, r, (size_t) V->A.n * sizeof (double));
   cycle (M, 0);
   memcpy (z, V->x, (size_t) V->A.n * sizeof (double));
}

/* ------------------------------------------------------------------ FGMRES (m) around the cycle */
static double dotp (int64_t n, const double *x, const double *y)
{
   double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
   for (int64_t i = 0; i < n; i++) acc += x[i] * y[i];
   return acc;
}

static void residual (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const double *x, const double *b, double *r)
{
#pragma omp parallel for schedule(static)
   for (int64_t i = 0; i < n; i++) {
      double acc = 0.0;
      for (int32_t e = rowptr[i]; e < rowptr[i + 1]; e++) acc += val[e] * x[colind[e]];
      r[i] = b ? b[i] - acc : acc;
   }
}

/* right-preconditioned flexible GMRES, one classical Gram-Schmidt pass (two with reorth), the inner iteration stops on the
 * recurrence's estimate and the true residual decides at every restart -- the driver of csrc/solver.hip.
 * returns 0 converged / 1 not converged / 2 breakdown */
ORA_EXPORT int ora_ml_fgmres (ora_ml *M, int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, int restart, int max_iters, double rtol,
                              int reorth, const double *b, double *x, int *iters_out, double *relres_out)
{
   const int m = restart;
   double *V = xmalloc ((size_t) n * (m + 1) * sizeof (double)), *Z = xmalloc ((size_t) n * m * sizeof (double)), *w = xmalloc ((size_t) n * sizeof (double));
   double *H = xcalloc ((size_t) (m + 1) * m, sizeof (double)), *cs = xmalloc (sizeof (double) * m), *sn = xmalloc (sizeof (double) * m);
   double *g = xmalloc (sizeof (double) * (m + 1)), *y = xmalloc (sizeof (double) * m), *h = xmalloc (sizeof (double) * (m + 2)), *h2 = xmalloc (sizeof (double) * (m + 2));
   int status = 1, its = 0;
   double relres = 0.0;
   const double bnorm = sqrt (dotp (n, b, b));
   memset (x, 0, (size_t) n * sizeof (double));
   if (!(bnorm > 0.0)) { status = 0; goto done; }
   const double target = rtol * bnorm;
   double inner_scale = 1.0;
   for (;;) {
      residual (n, rowptr, colind, val, x, b, w);
      const double beta = sqrt (dotp (n, w, w));
      relres = beta / bnorm;
      if (beta <= target) { status = 0; break; }
      if (its >= max_iters) { status = 1; break; }
      const double ib = 1.0 / beta;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) V[i] = ib * w[i];
      g[0] = beta;
      int j = 0, brk = 0;
      double est = beta;
      for (; j < m && its < max_iters; j++) {
         double *vj = V + (int64_t) j * n, *zj = Z + (int64_t) j * n;
         ora_ml_apply (M, vj, zj);
         residual (n, rowptr, colind, val, zj, NULL, w);
         for (int pass = 0; pass < (reorth ? 2 : 1); pass++) {
            double *hh = pass ? h2 : h;
            for (int i = 0; i <= j; i++) hh[i] = dotp (n, V + (int64_t) i * n, w);
#pragma omp parallel for schedule(static)
            for (int64_t q = 0; q < n; q++) {
               double a = w[q];
               for (int i = 0; i <= j; i++) a -= hh[i] * V[(int64_t) i * n + q];
               w[q] = a;
            }
         }
         if (reorth) for (int i = 0; i <= j; i++) h[i] += h2[i];
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
         est = fabs (g[j + 1]);
         if (est <= target * inner_scale || hj1 == 0.0) { j++; break; }
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
         residual (n, rowptr, colind, val, x, b, w);
         relres = sqrt (dotp (n, w, w)) / bnorm;
         status = relres * bnorm <= target ? 0 : 2;
         break;
      }
      /* a cycle that stopped on an estimate the true residual does not confirm: aim lower next time */
      residual (n, rowptr, colind, val, x, b, w);
      const double tr = sqrt (dotp (n, w, w));
      if (tr > target && est > 0.0 && est <= target * inner_scale) inner_scale = fmax (1e-3, fmin (inner_scale, 0.5 * est / tr));
   }
 done:
   if (iters_out) *iters_out = its;
   if (relres_out) *relres_out = relres;
   free (V); free (Z); free (w); free (H); free (cs); free (sn); free (g); free (y); free (h); free (h2);
   return status;
}
