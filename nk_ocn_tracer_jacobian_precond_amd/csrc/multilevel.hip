// Multilevel water-column preconditioner (NKP_PRECOND_MULTILEVEL).
//
// Why it exists: with exact water-column blocks alone, restarted GMRES needs >1e4 iterations
// on upwind3/centred Jacobians at 3 degrees and stalls outright at 1 degree (SURVEY.md section 7,
// hard part 1) -- which is why the reference uses a sparse direct solver
// (src/solve_ABglobal.c:353).  This preconditioner keeps the column-block kernel of colblock.hip
// as its smoother and adds the two things the block-Jacobi sweep lacks:
//
//  1. a monotone low-order twin L of A: every wrong-signed coupling BETWEEN water columns is
//     removed by symmetric artificial diffusion d_ij = max(0, -a_ij, -a_ji) (algebraic
//     upwinding: centred / upwind3 advection weights, src/matrix.c:1239-1273, 1610-1690, become
//     the donor-cell operator; the +-isopycnal cross terms, :881-930, become positive); entries
//     inside a column stay exact.  -L is an M-matrix, so column-block Gauss-Seidel converges on
//     it and on every Galerkin coarsening of it.
//  2. a hierarchy: columns are aggregated pairwise twice (~4 columns per aggregate, levels k
//     kept), P is piecewise constant, L_{l+1} = P^T L_l P; every level keeps the "contiguous
//     water column" layout, so the SAME wave-per-column kernels run on all levels.  Columns are
//     2-coloured and stored colour-major, so a Gauss-Seidel half-sweep is one row-range
//     residual SpMV + one block-range column solve.
//
// One V(nu,nu) cycle approximates L^-1; FGMRES (solver.hip) iterates on the original A.
// Setup is host code (O(nnz)); every cycle runs on the device.
#include "nkp_dev.h"
#include "multilevel.h"
#include "mlsetup.h"
#include "../../include/nkp.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <array>
#include <chrono>
#include <numeric>
#include <thread>
#include <vector>

namespace {

// allocator whose resize () leaves the new elements uninitialised: the setup's big arrays are written once by row-parallel
// loops right after they are sized, and a zero-fill by ONE thread (page faults included) was most of the "twin" time
template <class T>
struct RawAlloc {
   using value_type = T;
   RawAlloc () = default;
   template <class U> RawAlloc (const RawAlloc<U> &) {}
   T *allocate (size_t k) { return static_cast<T *> (::operator new (k * sizeof (T))); }
   void deallocate (T *q, size_t) { ::operator delete (q); }
   template <class U> void construct (U *q) noexcept { ::new ((void *) q) U; }                       // default-init: no store for int / double
   template <class U, class... A> void construct (U *q, A &&... a) { ::new ((void *) q) U (std::forward<A> (a)...); }
   template <class U> bool operator== (const RawAlloc<U> &) const { return true; }
   template <class U> bool operator!= (const RawAlloc<U> &) const { return false; }
};
using RawInts = std::vector<int, RawAlloc<int>>;
using RawDoubles = std::vector<double, RawAlloc<double>>;

struct HostCsr {
   int64_t n = 0;
   std::vector<int> rowptr;
   RawInts colind;
   RawDoubles val;
};

// ---------------------------------------------------------------- host threads for the setup loops
// host threads of the setup loops and the timing print of the aggregation: set at the start of every setup call from the
// solver's tuning (1 degree, 256-core host: Galerkin products 0.25 / 0.14 / 0.11 s with 16 / 32 / 64 threads; 32 leaves room
// for one process per GPU on an 8-GPU node)
thread_local int g_setup_threads = 0;
thread_local bool g_plan_times = false;

void use_setup_knobs (const nkp_tuning &t)
{
   g_setup_threads = t.setup_threads > 0 ? t.setup_threads : (int) std::min (32u, std::max (1u, std::thread::hardware_concurrency ()));
   g_plan_times = t.plan_times != 0;
}

int setup_threads () { return g_setup_threads > 0 ? g_setup_threads : 1; }

// run fn (chunk, first_row, last_row) on contiguous row chunks, one host thread each
template <class F>
void for_row_chunks (int64_t n, F fn)
{
   int nt = setup_threads ();
   if (n < 200000) nt = 1;
   std::vector<std::thread> pool;
   for (int c = 0; c < nt; c++) {
      const int64_t r0 = n * c / nt, r1 = n * (c + 1) / nt;
      if (nt == 1) fn (c, r0, r1);
      else pool.emplace_back ([=] () { fn (c, r0, r1); });
   }
   for (std::thread &th : pool) th.join ();
}

// ---------------------------------------------------------------- low-order twin
// L = A + D - diag(rowsum D), D_ij = max(0, -a_ij, -a_ji) for i, j in different columns.  Rows are sorted by column, so
// a_ji is found by bisection in row j (no transpose); three row-parallel passes: new values and kept-entry counts,
// prefix sum, fill.  Entries whose coupling becomes exactly zero are not stored.
void build_low_order (int64_t n, const int *rowptr, const int *colind, const double *val, const std::vector<int> &col_of, HostCsr &L)
{
   const int64_t nnz = rowptr[n];
   RawDoubles nv;
   nv.resize ((size_t) nnz);
   std::vector<int> keep ((size_t) n + 1, 0);
   for_row_chunks (n, [&] (int, int64_t r0, int64_t r1) {
      for (int64_t i = r0; i < r1; i++) {
         double dsum = 0.0;
         int diag_pos = -1, cnt = 0;
         for (int e = rowptr[i]; e < rowptr[i + 1]; e++) {
            const int j = colind[e];
            double a = val[e];
            if (j == i) { diag_pos = e; nv[e] = a; cnt++; continue; }
            if (col_of[j] != col_of[i]) {
               const int *lo = colind + rowptr[j], *hi = colind + rowptr[j + 1];
               const int *q = std::lower_bound (lo, hi, (int) i);
               const double aji = (q < hi && *q == (int) i) ? val[q - colind] : 0.0;
               double d = 0.0;
               if (-a > d) d = -a;
               if (-aji > d) d = -aji;
               a += d;
               dsum += d;
            }
            nv[e] = a;
            if (a != 0.0 || col_of[j] == col_of[i]) cnt++;          // in-column entries are always stored
         }
         if (diag_pos >= 0) nv[diag_pos] -= dsum;
         keep[(size_t) i + 1] = cnt;
      }
   });
   L.n = n;
   L.rowptr.assign ((size_t) n + 1, 0);
   for (int64_t i = 0; i < n; i++) L.rowptr[(size_t) i + 1] = L.rowptr[(size_t) i] + keep[(size_t) i + 1];
   L.colind.resize ((size_t) L.rowptr[(size_t) n]);
   L.val.resize ((size_t) L.rowptr[(size_t) n]);
   for_row_chunks (n, [&] (int, int64_t r0, int64_t r1) {
      for (int64_t i = r0; i < r1; i++) {
         int q = L.rowptr[(size_t) i];
         for (int e = rowptr[i]; e < rowptr[i + 1]; e++) {
            const int j = colind[e];
            if (j != i && col_of[j] != col_of[i] && nv[e] == 0.0) continue;
            L.colind[(size_t) q] = j;
            L.val[(size_t) q] = nv[e];
            q++;
         }
      }
   });
}

// ---------------------------------------------------------------- column graph helpers
struct ColGraph {
   std::vector<int> ptr, nbr;
   std::vector<double> w;
};

void build_col_graph (const HostCsr &L, const std::vector<int> &blk_start, const std::vector<int> &col_of, ColGraph &G)
{
   const int ncol = (int) blk_start.size () - 1;
   G.ptr.assign (ncol + 1, 0);
   G.nbr.clear ();
   G.w.clear ();
   std::vector<double> acc (ncol, 0.0);
   std::vector<int> touched;
   for (int c = 0; c < ncol; c++) {
      touched.clear ();
      for (int r = blk_start[c]; r < blk_start[c + 1]; r++)
         for (int e = L.rowptr[r]; e < L.rowptr[r + 1]; e++) {
            const int c2 = col_of[L.colind[e]];
            if (c2 == c) continue;
            if (acc[c2] == 0.0) touched.push_back (c2);
            acc[c2] += fabs (L.val[e]) + 1.0e-300;
         }
      std::sort (touched.begin (), touched.end ());
      for (int c2 : touched) {
         G.nbr.push_back (c2);
         G.w.push_back (acc[c2]);
         acc[c2] = 0.0;
      }
      G.ptr[c + 1] = (int) G.nbr.size ();
   }
}

// one pass of pairwise matching on a weighted graph; returns group id per node (ordered by first member)
int pairwise_match (int nn, const std::vector<int> &ptr, const std::vector<int> &nbr, const std::vector<double> &w, std::vector<int> &group)
{
   group.assign (nn, -1);
   int ng = 0;
   for (int c = 0; c < nn; c++) {
      if (group[c] >= 0) continue;
      int best = -1;
      double bw = 0.0;
      for (int q = ptr[c]; q < ptr[c + 1]; q++)
         if (group[nbr[q]] < 0 && nbr[q] != c && w[q] > bw) { bw = w[q]; best = nbr[q]; }
      group[c] = ng;
      if (best >= 0) group[best] = ng;
      ng++;
   }
   return ng;
}

// collapse a node graph onto groups
void collapse_graph (int nn, int ng, const std::vector<int> &group, const ColGraph &G, ColGraph &H)
{
   std::vector<std::vector<int>> members (ng);
   for (int c = 0; c < nn; c++) members[group[c]].push_back (c);
   H.ptr.assign (ng + 1, 0);
   H.nbr.clear ();
   H.w.clear ();
   std::vector<double> acc (ng, 0.0);
   std::vector<int> touched;
   for (int g = 0; g < ng; g++) {
      touched.clear ();
      for (int c : members[g])
         for (int q = G.ptr[c]; q < G.ptr[c + 1]; q++) {
            const int g2 = group[G.nbr[q]];
            if (g2 == g) continue;
            if (acc[g2] == 0.0) touched.push_back (g2);
            acc[g2] += G.w[q];
         }
      std::sort (touched.begin (), touched.end ());
      for (int g2 : touched) {
         H.nbr.push_back (g2);
         H.w.push_back (acc[g2]);
         acc[g2] = 0.0;
      }
      H.ptr[g + 1] = (int) H.nbr.size ();
   }
}

// greedy 2-colouring: each column takes the colour its already-coloured neighbours use least (by weight)
void two_colour (int ncol, const ColGraph &G, std::vector<int> &colour)
{
   colour.assign (ncol, -1);
   for (int c = 0; c < ncol; c++) {
      double w0 = 0.0, w1 = 0.0;
      for (int q = G.ptr[c]; q < G.ptr[c + 1]; q++) {
         const int k = colour[G.nbr[q]];
         if (k == 0) w0 += G.w[q];
         else if (k == 1) w1 += G.w[q];
      }
      colour[c] = (w0 <= w1) ? 0 : 1;
      if (w0 == 0.0 && w1 == 0.0) colour[c] = 0;
   }
}


// ---------------------------------------------------------------- split aggregates (geometric groups, connectivity-aware)
// A group of columns (2 x 2 or 4 x 4 in (i, j)) is NOT turned into one coarse column blindly: at every depth k the
// members that are wet at k form one coarse cell per CONNECTED set (lateral couplings of the level operator between
// members), because a piecewise-constant cell over mutually uncoupled water (two sides of a ridge, a deep pocket
// next to open water) cannot represent the near-kernel of the operator -- it is constant per connected piece, not per
// group -- and neither the column smoother nor any coarser level then removes that error (measured: the two-grid
// iteration with an exact coarse solve needs 88 Krylov steps at a 0.25-degree cell Courant number, 21 with the split).
//  * same-depth connected sets of at most `pocket` cells are merged into one coarse cell even across groups (a deep
//    pocket is a strongly coupled cluster hanging on weak vertical diffusion: it must become ONE unknown);
//  * the sets are threaded through depth into coarse columns: the child set with the largest overlap continues its
//    parent's column, every other child starts a stub column (first depth > 0) at the same (i, j);
//  * a stub of the fine level that no outside row feels (every coupling into it is < tau x that row's diagonal) is a
//    leaf: the column solve makes it follow its neighbours exactly, so it is absorbed into the coarse cell it hangs
//    from instead of surviving as an unknown on every coarser level.
// Rows keep their depth: row r of column c sits at depth ktop[c] + (r - blk_start[c]).
struct SplitResult {
   std::vector<int> cmap;                     // fine row -> coarse row
   std::vector<int> blk_start, ktop, gi, gj, gt;   // coarse columns
   int absorbed = 0, stubs = 0;
};

// Lock-free union-find (several host threads unite concurrently): a root is only ever linked under a LOWER index with a
// compare-and-swap, so the final root of every set is its lowest row whatever the interleaving -- the partition and the
// numbering derived from it are deterministic.
struct UnionFind {
   std::unique_ptr<std::atomic<int>[]> p;
   size_t n;
   explicit UnionFind (size_t n_) : p (n_ ? new std::atomic<int>[n_] : nullptr), n (n_)
   {
      for (size_t i = 0; i < n; i++) p[i].store ((int) i, std::memory_order_relaxed);
   }
   int find (int x)
   {
      for (;;) {
         const int px = p[x].load (std::memory_order_relaxed);
         if (px == x) return x;
         const int gp = p[px].load (std::memory_order_relaxed);
         if (gp != px) { int expect = px; p[x].compare_exchange_weak (expect, gp, std::memory_order_relaxed); }   // path halving
         x = px;
      }
   }
   void unite (int a, int b)
   {
      for (;;) {
         a = find (a);
         b = find (b);
         if (a == b) return;
         if (a > b) std::swap (a, b);                              // link the higher root b under the lower root a
         int expect = b;
         if (p[b].compare_exchange_strong (expect, a, std::memory_order_relaxed)) return;
      }
   }
};

void split_aggregate (const HostCsr &L, const std::vector<int> &blk_start, const std::vector<int> &col_of, const std::vector<int> &ktop,
                      const std::vector<int> &group, const std::vector<int> &ggi, const std::vector<int> &ggj, const std::vector<int> &ggt,
                      const std::vector<int> &col_t, int pocket, double theta, double tau, SplitResult &R)
{
   const int64_t n = L.n;
   const int ncol = (int) blk_start.size () - 1;
   const bool timing = g_plan_times;
   auto tick0 = std::chrono::steady_clock::now ();
   auto lap = [&] (const char *what) {
      if (!timing) return;
      auto now = std::chrono::steady_clock::now ();
      printf ("   split_aggregate (%lld rows): %-28s %.3f s\n", (long long) n, what, std::chrono::duration<double> (now - tick0).count ());
      tick0 = now;
   };
   auto depth = [&] (int r) { const int c = col_of[r]; return ktop[c] + (r - blk_start[c]); };
   auto row_at = [&] (int c, int k) -> int { const int r = blk_start[c] + (k - ktop[c]); return (k >= ktop[c] && r < blk_start[c + 1]) ? r : -1; };
   // per row: strongest lateral coupling (only needed for a threshold theta > 0); per column: how strongly any outside
   // row of the same tracer feels it, and its own strongest coupling (only needed where stub columns exist)
   bool have_stubs = false;
   for (int c = 0; c < ncol && !have_stubs; c++) have_stubs = ktop[c] > 0;
   std::vector<double> diag, rowmax, felt, best;
   std::vector<int> anchor (ncol, -1);
   std::vector<char> dang (ncol, 0);
   if (theta > 0.0) {
      rowmax.assign (n, 0.0);
      for (int64_t r = 0; r < n; r++)
         for (int e = L.rowptr[r]; e < L.rowptr[r + 1]; e++) {
            const int c2 = col_of[L.colind[e]];
            if (c2 != col_of[r] && col_t[c2] == col_t[col_of[r]]) rowmax[r] = std::max (rowmax[r], fabs (L.val[e]));
         }
   }
   if (have_stubs && tau > 0.0) {
      diag.assign (n, 0.0);
      felt.assign (ncol, 0.0);
      best.assign (ncol, -1.0);
      for (int64_t r = 0; r < n; r++)
         for (int e = L.rowptr[r]; e < L.rowptr[r + 1]; e++)
            if (L.colind[e] == r) diag[r] = fabs (L.val[e]);
      for (int64_t r = 0; r < n; r++) {
         const int c = col_of[r];
         for (int e = L.rowptr[r]; e < L.rowptr[r + 1]; e++) {
            const int j = L.colind[e], c2 = col_of[j];
            if (c2 == c || col_t[c2] != col_t[c]) continue;
            if (ktop[c2] == 0 && ktop[c] == 0) continue;                // neither end is a stub
            const double v = fabs (L.val[e]);
            const double f = diag[r] > 0.0 ? v / diag[r] : 1.0e300;
            if (f > felt[c2]) felt[c2] = f;
            if (v >= best[c]) { best[c] = v; anchor[c] = j; }          // strongest coupling of the column, ties -> later entry
         }
      }
      for (int c = 0; c < ncol; c++) dang[c] = (ktop[c] > 0 && felt[c] < tau && anchor[c] >= 0);
      std::vector<char> bad (ncol, 0);
      for (int c = 0; c < ncol; c++) bad[c] = dang[c] && dang[col_of[anchor[c]]];
      for (int c = 0; c < ncol; c++) if (bad[c]) dang[c] = 0;
   }
   lap ("stub analysis");
   // lateral edges between cells of the same depth, united on the fly: U0 over all of them (it finds the small same-depth
   // sets = pockets), U over the edges inside a group; a second scan of the pockets' rows adds their cross-group edges to U
   UnionFind U (n), U0 (pocket > 0 ? n : 0);
   auto scan_row = [&] (int64_t r, auto &&visit) {
      const int c = col_of[r];
      if (dang[c]) return;
      const int k = depth ((int) r);
      for (int e = L.rowptr[r]; e < L.rowptr[r + 1]; e++) {
         const int j = L.colind[e], c2 = col_of[j];
         if (c2 == c || dang[c2] || col_t[c2] != col_t[c]) continue;
         const int dk = depth (j) - k;
         if (dk < -1 || dk > 1) continue;
         if (theta > 0.0 && fabs (L.val[e]) < theta * rowmax[r]) continue;
         const int t = row_at (c2, k);
         if (t >= 0) visit ((int) r, t, group[c] == group[c2]);
      }
   };
   for_row_chunks (n, [&] (int, int64_t r0, int64_t r1) {
      for (int64_t r = r0; r < r1; r++)
         scan_row (r, [&] (int a, int b, bool same) {
            if (pocket > 0) U0.unite (a, b);
            if (same) U.unite (a, b);
         });
   });
   if (pocket > 0) {
      std::vector<int> root0 (n), size (n, 0);
      for_row_chunks (n, [&] (int, int64_t r0, int64_t r1) { for (int64_t r = r0; r < r1; r++) root0[r] = U0.find ((int) r); });
      for (int64_t r = 0; r < n; r++) size[root0[r]]++;
      for_row_chunks (n, [&] (int, int64_t r0, int64_t r1) {
         for (int64_t r = r0; r < r1; r++)
            if (size[root0[r]] <= pocket && size[root0[r]] > 1)
               scan_row (r, [&] (int a, int b, bool same) { if (!same) U.unite (a, b); });
      });
   }
   lap ("union-find over the edges");
   // components numbered in order of their lowest row
   std::vector<int> comp (n, -1);
   int ncomp = 0;
   for (int64_t r = 0; r < n; r++) {
      const int root = U.find ((int) r);
      if (comp[root] < 0) comp[root] = ncomp++;       // root is the lowest row of its set, so it is met first
      comp[r] = comp[root];
   }
   std::vector<int> kcomp (ncomp, 0);
   for (int64_t r = 0; r < n; r++) kcomp[comp[r]] = depth ((int) r);
   lap ("component numbering");
   // overlaps between a set and the sets directly below it: the pairs (set of row r, set of the row below r) are bucketed
   // by parent with a counting sort (set ids are dense), every bucket -- the handful of rows of one set -- is sorted and
   // its runs counted
   std::vector<int> bestpar (ncomp, -1), bestpar_cnt (ncomp, 0), bestchi (ncomp, -1), bestchi_cnt (ncomp, 0);
   {
      std::vector<int> bptr ((size_t) ncomp + 1, 0);
      for (int c = 0; c < ncol; c++)
         for (int r = blk_start[c]; r + 1 < blk_start[c + 1]; r++) bptr[(size_t) comp[r] + 1]++;
      for (int q = 0; q < ncomp; q++) bptr[(size_t) q + 1] += bptr[(size_t) q];
      std::vector<int> child ((size_t) bptr[(size_t) ncomp]);
      {
         std::vector<int> fill (bptr.begin (), bptr.end () - 1);
         for (int c = 0; c < ncol; c++)
            for (int r = blk_start[c]; r + 1 < blk_start[c + 1]; r++) child[(size_t) fill[(size_t) comp[r]]++] = comp[r + 1];
      }
      // parents in ascending id, children ascending inside a bucket: a strict '>' keeps the lowest id on ties, like the
      // sorted list of pairs did
      for (int par = 0; par < ncomp; par++) {
         int *b0 = child.data () + bptr[(size_t) par], *b1 = child.data () + bptr[(size_t) par + 1];
         if (b1 - b0 > 1) std::sort (b0, b1);
         for (int *q = b0; q < b1;) {
            int *q2 = q;
            while (q2 < b1 && *q2 == *q) q2++;
            const int chi = *q, cnt = (int) (q2 - q);
            if (cnt > bestpar_cnt[chi]) { bestpar_cnt[chi] = cnt; bestpar[chi] = par; }
            if (cnt > bestchi_cnt[par]) { bestchi_cnt[par] = cnt; bestchi[par] = chi; }
            q = q2;
         }
      }
   }
   lap ("overlap pairs (buckets)");
   // coarse columns: sets in order of depth (counting sort), then of id
   std::vector<int> order (ncomp);
   {
      int kmax = 0;
      for (int q = 0; q < ncomp; q++) kmax = std::max (kmax, kcomp[q]);
      std::vector<int> kptr ((size_t) kmax + 2, 0);
      for (int q = 0; q < ncomp; q++) kptr[(size_t) kcomp[q] + 1]++;
      for (int k = 0; k <= kmax; k++) kptr[(size_t) k + 1] += kptr[(size_t) k];
      for (int q = 0; q < ncomp; q++) order[(size_t) kptr[(size_t) kcomp[q]]++] = q;
   }
   std::vector<int> ccol (ncomp, -1), cc_ktop, cc_len;
   for (int id : order) {
      const int par = bestpar[id];
      if (par >= 0 && bestchi[par] == id) {
         ccol[id] = ccol[par];
         cc_len[ccol[id]]++;
      } else {
         ccol[id] = (int) cc_ktop.size ();
         cc_ktop.push_back (kcomp[id]);
         cc_len.push_back (1);
      }
   }
   lap ("threading");
   // absorbed stubs own no coarse column: drop the (now empty) columns their sets opened
   const int nraw = (int) cc_ktop.size ();
   // a coarse column sits at the (i, j) of the group of its lowest fine row (a merged pocket can span groups)
   std::vector<char> used (nraw, 0);
   std::vector<int> cc_group (nraw, 0);
   for (int64_t r = 0; r < n; r++)
      if (!dang[col_of[r]] && !used[ccol[comp[r]]]) { used[ccol[comp[r]]] = 1; cc_group[ccol[comp[r]]] = group[col_of[r]]; }
   std::vector<int> newid (nraw, -1);
   int ncc = 0;
   for (int q = 0; q < nraw; q++) if (used[q]) newid[q] = ncc++;
   R.blk_start.assign (ncc + 1, 0);
   R.ktop.resize (ncc); R.gi.resize (ncc); R.gj.resize (ncc); R.gt.resize (ncc);
   for (int q = 0; q < nraw; q++) {
      if (!used[q]) continue;
      const int a = newid[q];
      R.blk_start[a + 1] = cc_len[q];
      R.ktop[a] = cc_ktop[q];
      R.gi[a] = ggi[cc_group[q]]; R.gj[a] = ggj[cc_group[q]]; R.gt[a] = ggt[cc_group[q]];
      if (cc_ktop[q] > 0) R.stubs++;
   }
   for (int a = 0; a < ncc; a++) R.blk_start[a + 1] += R.blk_start[a];
   R.cmap.assign (n, -1);
   for (int64_t r = 0; r < n; r++) {
      if (dang[col_of[r]]) continue;
      const int a = newid[ccol[comp[r]]];
      R.cmap[r] = R.blk_start[a] + (depth ((int) r) - R.ktop[a]);
   }
   lap ("coarse columns and map");
   for (int c = 0; c < ncol; c++) {
      if (!dang[c]) continue;
      R.absorbed++;
      const int target = R.cmap[anchor[c]];
      for (int r = blk_start[c]; r < blk_start[c + 1]; r++) R.cmap[r] = target;
   }
}

// Galerkin product with a piecewise-constant P given as fine row -> coarse row
void galerkin (const HostCsr &L, const std::vector<int> &cmap, int64_t nc, HostCsr &C)
{
   // coarse row -> fine rows
   std::vector<int> rptr (nc + 1, 0), ridx (L.n);
   for (int64_t i = 0; i < L.n; i++) rptr[cmap[i] + 1]++;
   for (int64_t I = 0; I < nc; I++) rptr[I + 1] += rptr[I];
   {
      std::vector<int> fill (rptr.begin (), rptr.end () - 1);
      for (int64_t i = 0; i < L.n; i++) ridx[fill[cmap[i]]++] = (int) i;
   }
   // coarse rows in parallel: every thread owns a contiguous run of coarse rows, with its own accumulator over the
   // coarse columns, and appends to its own output; the pieces are stitched together in row order
   C.n = nc;
   C.rowptr.assign (nc + 1, 0);
   const int nt_max = setup_threads ();
   std::vector<std::vector<int>> pc (nt_max);
   std::vector<std::vector<double>> pv (nt_max);
   std::vector<int64_t> first (nt_max, 0), last (nt_max, 0);
   for_row_chunks (nc, [&] (int t, int64_t I0, int64_t I1) {
      first[t] = I0;
      last[t] = I1;
      std::vector<double> acc (nc, 0.0);
      std::vector<char> mark (nc, 0);
      std::vector<int> touched;
      std::vector<int> &oc = pc[t];
      std::vector<double> &ov = pv[t];
      oc.reserve ((size_t) ((L.colind.size () / 2) * (double) (I1 - I0) / (double) (nc ? nc : 1)) + 16);
      ov.reserve (oc.capacity ());
      for (int64_t I = I0; I < I1; I++) {
         touched.clear ();
         for (int q = rptr[I]; q < rptr[I + 1]; q++) {
            const int i = ridx[q];
            for (int e = L.rowptr[i]; e < L.rowptr[i + 1]; e++) {
               const int J = cmap[L.colind[e]];
               if (!mark[J]) { mark[J] = 1; touched.push_back (J); }
               acc[J] += L.val[e];
            }
         }
         std::sort (touched.begin (), touched.end ());
         int cnt = 0;
         for (int J : touched) {
            if (acc[J] != 0.0 || J == I) {
               oc.push_back (J);
               ov.push_back (acc[J]);
               cnt++;
            }
            acc[J] = 0.0;
            mark[J] = 0;
         }
         C.rowptr[I + 1] = cnt;
      }
   });
   for (int64_t I = 0; I < nc; I++) C.rowptr[I + 1] += C.rowptr[I];
   C.colind.resize ((size_t) C.rowptr[nc]);
   C.val.resize ((size_t) C.rowptr[nc]);
   {
      // every piece into its place, one thread per piece (the destination pages are first touched here)
      std::vector<std::thread> pool;
      for (int t = 0; t < nt_max; t++) {
         if (last[t] <= first[t]) continue;
         pool.emplace_back ([&, t] () {
            std::copy (pc[t].begin (), pc[t].end (), C.colind.begin () + C.rowptr[first[t]]);
            std::copy (pv[t].begin (), pv[t].end (), C.val.begin () + C.rowptr[first[t]]);
         });
      }
      for (std::thread &th : pool) th.join ();
   }
}

// dense inverse by Gauss-Jordan with partial pivoting (coarsest level only); returns false if singular
bool dense_inverse (int n, std::vector<double> &a /* row-major n*n, overwritten by its inverse */)
{
   std::vector<double> inv ((size_t) n * n, 0.0);
   for (int i = 0; i < n; i++) inv[(size_t) i * n + i] = 1.0;
   for (int k = 0; k < n; k++) {
      int p = k;
      double mx = fabs (a[(size_t) k * n + k]);
      for (int i = k + 1; i < n; i++)
         if (fabs (a[(size_t) i * n + k]) > mx) { mx = fabs (a[(size_t) i * n + k]); p = i; }
      if (!(mx > 0.0)) return false;
      if (p != k)
         for (int c = 0; c < n; c++) {
            std::swap (a[(size_t) k * n + c], a[(size_t) p * n + c]);
            std::swap (inv[(size_t) k * n + c], inv[(size_t) p * n + c]);
         }
      const double piv = 1.0 / a[(size_t) k * n + k];
      for (int c = 0; c < n; c++) { a[(size_t) k * n + c] *= piv; inv[(size_t) k * n + c] *= piv; }
      for (int i = 0; i < n; i++) {
         if (i == k) continue;
         const double f = a[(size_t) i * n + k];
         if (f == 0.0) continue;
         double *ai = &a[(size_t) i * n], *ak = &a[(size_t) k * n], *ii = &inv[(size_t) i * n], *ik = &inv[(size_t) k * n];
         for (int c = 0; c < n; c++) { ai[c] -= f * ak[c]; ii[c] -= f * ik[c]; }
      }
   }
   a.swap (inv);
   return true;
}


// ---------------------------------------------------------------- dense inverse on the device (coarsest level)
// Gauss-Jordan with partial pivoting, one elimination step = four small launches; the same operations as the host routine
// above (swap, scale the pivot row by the reciprocal, subtract f x pivot row from every other row), every element updated
// by one multiply and one subtract, so the result has the same bits -- and a 1450-row inverse takes 40 ms instead of the
// second it cost on the host (a third of the whole 1 degree setup).
__global__ void gj_pivot_kernel (const double *__restrict__ a, int n, int k, int *__restrict__ piv, double *__restrict__ pivval)
{
   __shared__ double smax[256];
   __shared__ int sidx[256];
   double mx = -1.0;
   int p = k;
   for (int i = k + (int) threadIdx.x; i < n; i += 256) {
      const double v = fabs (a[(size_t) i * n + k]);
      if (v > mx) { mx = v; p = i; }               // ascending i per thread: the first maximum wins
   }
   smax[threadIdx.x] = mx;
   sidx[threadIdx.x] = p;
   __syncthreads ();
   for (int off = 128; off > 0; off >>= 1) {
      if ((int) threadIdx.x < off) {
         const double o = smax[threadIdx.x + off];
         const int oi = sidx[threadIdx.x + off];
         if (o > smax[threadIdx.x] || (o == smax[threadIdx.x] && oi < sidx[threadIdx.x])) { smax[threadIdx.x] = o; sidx[threadIdx.x] = oi; }
      }
      __syncthreads ();
   }
   if (threadIdx.x == 0) {
      piv[0] = sidx[0];
      if (!(smax[0] > 0.0)) piv[1] = 1;            // singular
      pivval[0] = a[(size_t) sidx[0] * n + k];
   }
}

__global__ void gj_swap_scale_kernel (double *__restrict__ a, double *__restrict__ inv, int n, int k, const int *__restrict__ piv, const double *__restrict__ pivval)
{
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c >= n) return;
   const int p = piv[0];
   const double r = 1.0 / pivval[0];
   double *m[2] = { a, inv };
   for (int w = 0; w < 2; w++) {
      const double vk = m[w][(size_t) k * n + c], vp = m[w][(size_t) p * n + c];
      m[w][(size_t) k * n + c] = vp * r;
      if (p != k) m[w][(size_t) p * n + c] = vk;
   }
}

__global__ void gj_column_kernel (const double *__restrict__ a, int n, int k, double *__restrict__ fcol)
{
   const int i = blockIdx.x * 256 + threadIdx.x;
   if (i < n) fcol[i] = a[(size_t) i * n + k];
}

__global__ void gj_update_kernel (double *__restrict__ a, double *__restrict__ inv, int n, int k, const double *__restrict__ fcol)
{
   const int i = blockIdx.y;
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c >= n || i == k) return;
   const double f = fcol[i];
   if (f == 0.0) return;
   a[(size_t) i * n + c] -= f * a[(size_t) k * n + c];
   inv[(size_t) i * n + c] -= f * inv[(size_t) k * n + c];
}

// a (device, row-major n x n) is destroyed, *inv_out receives a device buffer with the inverse; false if singular / no memory
bool dense_inverse_device (int n, const std::vector<double> &host_a, double **inv_out, size_t *bytes, hipStream_t st)
{
   double *a = nullptr, *inv = nullptr, *fcol = nullptr, *pivval = nullptr;
   int *piv = nullptr;
   const size_t nn = (size_t) n * n;
   bool ok = hipMalloc ((void **) &a, nn * sizeof (double)) == hipSuccess && hipMalloc ((void **) &inv, nn * sizeof (double)) == hipSuccess &&
             hipMalloc ((void **) &fcol, (size_t) n * sizeof (double)) == hipSuccess && hipMalloc ((void **) &pivval, sizeof (double)) == hipSuccess &&
             hipMalloc ((void **) &piv, 2 * sizeof (int)) == hipSuccess;
   if (ok) {
      std::vector<double> eye (nn, 0.0);
      for (int i = 0; i < n; i++) eye[(size_t) i * n + i] = 1.0;
      ok = hipMemcpy (a, host_a.data (), nn * sizeof (double), hipMemcpyHostToDevice) == hipSuccess &&
           hipMemcpy (inv, eye.data (), nn * sizeof (double), hipMemcpyHostToDevice) == hipSuccess && hipMemset (piv, 0, 2 * sizeof (int)) == hipSuccess;
   }
   if (ok) {
      const int cb = (n + 255) / 256;
      for (int k = 0; k < n; k++) {
         hipLaunchKernelGGL (gj_pivot_kernel, dim3 (1), dim3 (256), 0, st, a, n, k, piv, pivval);
         hipLaunchKernelGGL (gj_swap_scale_kernel, dim3 (cb), dim3 (256), 0, st, a, inv, n, k, piv, pivval);
         hipLaunchKernelGGL (gj_column_kernel, dim3 (cb), dim3 (256), 0, st, a, n, k, fcol);
         hipLaunchKernelGGL (gj_update_kernel, dim3 (cb, n), dim3 (256), 0, st, a, inv, n, k, fcol);
      }
      int flags[2] = { 0, 0 };
      ok = hipMemcpyAsync (flags, piv, sizeof flags, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize (st) == hipSuccess && flags[1] == 0;
   }
   if (a) (void) hipFree (a);
   if (fcol) (void) hipFree (fcol);
   if (pivval) (void) hipFree (pivval);
   if (piv) (void) hipFree (piv);
   if (!ok) { if (inv) (void) hipFree (inv); return false; }
   *inv_out = inv;
   *bytes += nn * sizeof (double);
   return true;
}

template <class T>
bool upload (T **dst, const T *src, size_t count, size_t *bytes)
{
   void *q = nullptr;
   const size_t b = (count ? count : 1) * sizeof (T);
   if (hipMalloc (&q, b) != hipSuccess) return false;
   if (count && src && hipMemcpy (q, src, count * sizeof (T), hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree (q); return false; }
   *dst = (T *) q;
   *bytes += b;
   return true;
}

// same with `pad` zeroed extra elements (the SpMV reads matrix entries in aligned pairs)
template <class T>
bool upload_padded (T **dst, const T *src, size_t count, size_t pad, size_t *bytes)
{
   void *q = nullptr;
   const size_t b = (count + pad) * sizeof (T);
   if (hipMalloc (&q, b) != hipSuccess) return false;
   if (hipMemset (q, 0, b) != hipSuccess) { (void) hipFree (q); return false; }
   if (count && hipMemcpy (q, src, count * sizeof (T), hipMemcpyHostToDevice) != hipSuccess) { (void) hipFree (q); return false; }
   *dst = (T *) q;
   *bytes += b;
   return true;
}

// ---------------------------------------------------------------- natural-order data of every level (host)
struct Nat {
   HostCsr L;
   std::vector<int> blk_start, col_of, colour, agg;   // per column: colour, aggregate id
   std::vector<int> cmap;                            // fine row -> coarse row (natural orders)
   std::vector<int> perm, inv;                       // perm[new] = old ; inv[old] = new  (colour-major)
   std::vector<int> gi, gj, gt;                      // optional grid position / tracer of every column
   std::vector<int> ktop;                            // depth of the first row of every column (0 except for stub columns)
   int nagg = 0;
   int ncol0 = 0;                                    // columns of colour 0
};

struct SetupTimes { double low = 0.0, graph = 0.0, galerkin = 0.0; };

// knobs of the hierarchy construction (nkp_tuning; defaults are the measured best)
struct PlanKnobs {
   int split = 1, pocket = 4, big_from = -3, huge_from = -1;
   double theta = 0.0, tau = 0.01;
};

PlanKnobs plan_knobs (const nkp_tuning &t)
{
   PlanKnobs k;
   k.split = t.ml_split != 0;
   k.pocket = t.ml_pocket;
   k.theta = t.ml_theta;
   k.tau = t.ml_tau;
   k.big_from = t.ml_big_from;
   k.huge_from = t.ml_huge_from;
   return k;
}

// level-0 column arrays (and, with_twin, the low-order twin of A on the host)
void init_first_nat (Nat &N, int64_t n, const int *rowptr, const int *colind, const double *val, const int *blk_start_in, int64_t nblk,
                     const int *col_i, const int *col_j, const int *col_t, int tracer_cnt, bool with_twin, SetupTimes &T)
{
   using clk = std::chrono::steady_clock;
   N.blk_start.assign (blk_start_in, blk_start_in + nblk + 1);
   N.ktop.assign (nblk, 0);
   if (with_twin) {
      N.col_of.resize (n);
      for (int64_t c = 0; c < nblk; c++)
         for (int r = N.blk_start[c]; r < N.blk_start[c + 1]; r++) N.col_of[r] = (int) c;
      auto t0 = clk::now ();
      build_low_order (n, rowptr, colind, val, N.col_of, N.L);
      T.low += std::chrono::duration<double> (clk::now () - t0).count ();
   }
   if (col_i && col_j) {
      N.gi.assign (col_i, col_i + nblk);
      N.gj.assign (col_j, col_j + nblk);
      N.gt.resize (nblk);
      const int64_t per = (tracer_cnt > 1 && nblk % tracer_cnt == 0) ? nblk / tracer_cnt : nblk;
      // tracer of a column: positional (tracer-major rows, src/matrix.c:778-784) unless the caller names it -- the
      // distributed flavour appends the neighbouring ranks' overlap columns behind its own
      for (int64_t c = 0; c < nblk; c++) N.gt[c] = col_t ? col_t[c] : (int) (c / per);
   }
}

// 2 x 2 blocks of columns in (i, j) (4 x 4 with sh = 2), never across tracers; group ids in order of first member.
// Returns the number of groups; agg[c] = group of column c, cgi / cgj / cgt = position and tracer of every group.
int geo_groups (const Nat &N, int sh, std::vector<int> &agg, std::vector<int> &cgi, std::vector<int> &cgj, std::vector<int> &cgt)
{
   const int ncol = (int) N.blk_start.size () - 1;
   std::vector<std::pair<std::array<int, 3>, int>> sorted (ncol);
   for (int c = 0; c < ncol; c++) sorted[c] = { { N.gt[c], N.gj[c] >> sh, N.gi[c] >> sh }, c };
   std::sort (sorted.begin (), sorted.end ());
   std::vector<int> gid_sorted (ncol), first_member;
   int ng = 0;
   for (int q = 0; q < ncol; q++) {
      if (q == 0 || sorted[q].first != sorted[q - 1].first) { first_member.push_back (sorted[q].second); ng++; }
      gid_sorted[sorted[q].second] = ng - 1;
   }
   // renumber groups by their first (lowest natural index) member so coarse columns keep the j, i order
   std::vector<int> order (ng);
   std::iota (order.begin (), order.end (), 0);
   std::sort (order.begin (), order.end (), [&] (int a, int b) { return first_member[a] < first_member[b]; });
   std::vector<int> newid (ng);
   for (int q = 0; q < ng; q++) newid[order[q]] = q;
   agg.resize (ncol);
   cgi.resize (ng); cgj.resize (ng); cgt.resize (ng);
   for (int c = 0; c < ncol; c++) {
      const int a = newid[gid_sorted[c]];
      agg[c] = a;
      cgi[a] = N.gi[c] >> sh; cgj[a] = N.gj[c] >> sh; cgt[a] = N.gt[c];
   }
   return ng;
}

// 2 x 2 groups on the big levels, 4 x 4 from level 3 down: every kernel of a small level runs at its latency
// floor, so fewer small levels pay (1 degree: 8 -> 6 levels, +5 % iterations, -14 % cycle time);
// NKP_ML_BIG_FROM=l moves the switch, -1 disables it (from level 2 it costs +68 % iterations)
// (round 2, with the connectivity-aware cells and omega = 1.1: grids of fewer than 200 000 columns per tracer
// keep the switch at level 3 -- 1 degree: 64 iterations / 0.21 s either way -- larger grids coarsen 2 x 2 all the
// way, where the better hierarchy outweighs two more latency-bound levels: 0.5 degree 92 -> 79 iterations,
// 0.95 -> 0.81 s; 0.25 degree 129 -> 105, 4.1 -> 3.4 s)
int group_shift (const PlanKnobs &K, int level, int ncol_level0, int tracer_cnt)
{
   int bf = K.big_from;
   if (bf == -3) bf = ncol_level0 / (tracer_cnt > 0 ? tracer_cnt : 1) >= 200000 ? -1 : 3;
   if (K.huge_from >= 0 && level >= K.huge_from) return 3;          // 8 x 8 groups (A/B knob ml_huge_from)
   return (bf >= 0 && level >= bf) ? 2 : 1;
}

// colour of every column and the colour-major row order: newstart[c] = first row of column c in that order
void colour_major_columns (Nat &N, const std::vector<int> &colour, std::vector<int> &newstart, std::vector<int> &pblk)
{
   const int ncol = (int) N.blk_start.size () - 1;
   newstart.assign (ncol, 0);
   pblk.clear ();
   pblk.reserve (ncol + 1);
   pblk.push_back (0);
   N.ncol0 = 0;
   for (int pass = 0; pass < 2; pass++)
      for (int c = 0; c < ncol; c++)
         if (colour[c] == pass) {
            if (pass == 0) N.ncol0++;
            newstart[c] = pblk.back ();
            pblk.push_back (pblk.back () + (N.blk_start[c + 1] - N.blk_start[c]));
         }
}

// colouring, aggregation and Galerkin product of every level from nat.back () on (which holds its operator); host only
// (no HIP call).  level0 = index of nat[0] in the whole hierarchy (levels above it were built on the device).
void extend_nat_levels (std::vector<Nat> &nat, int level0, int ncol_level0, int tracer_cnt, int max_levels, int coarsest_rows, int verbose, int rank,
                        const PlanKnobs &K, SetupTimes &T)
{
   using clk = std::chrono::steady_clock;
   auto secs = [] (clk::time_point a) { return std::chrono::duration<double> (clk::now () - a).count (); };
   double &t_graph = T.graph, &t_galerkin = T.galerkin;
   for (int l = (int) nat.size () - 1;; l++) {
      Nat &N = nat[l];
      const int ncol = (int) N.blk_start.size () - 1;
      ColGraph G;
      const bool geo = !N.gi.empty ();
      if (!geo) { auto t0 = clk::now (); build_col_graph (N.L, N.blk_start, N.col_of, G); t_graph += secs (t0); }
      if (geo) {
         N.colour.resize (ncol);
         for (int c = 0; c < ncol; c++) N.colour[c] = (N.gi[c] + N.gj[c]) & 1;
      } else
         two_colour (ncol, G, N.colour);
      // colour-major permutation of rows
      N.perm.clear ();
      N.perm.reserve (N.L.n);
      N.ncol0 = 0;
      for (int pass = 0; pass < 2; pass++)
         for (int c = 0; c < ncol; c++)
            if (N.colour[c] == pass) {
               if (pass == 0) N.ncol0++;
               for (int r = N.blk_start[c]; r < N.blk_start[c + 1]; r++) N.perm.push_back (r);
            }
      N.inv.resize (N.L.n);
      for (int64_t i = 0; i < N.L.n; i++) N.inv[N.perm[i]] = (int) i;

      const bool last = (level0 + l + 1 >= max_levels) || (N.L.n <= coarsest_rows) || ncol <= 4;
      if (last) break;
      int n2 = 0;
      N.agg.resize (ncol);
      std::vector<int> cgi, cgj, cgt;
      if (geo) {
         n2 = geo_groups (N, group_shift (K, level0 + l, ncol_level0, tracer_cnt), N.agg, cgi, cgj, cgt);
      } else {
         // two passes of pairwise matching -> aggregates of up to 4 columns
         std::vector<int> g1, g2;
         const int n1 = pairwise_match (ncol, G.ptr, G.nbr, G.w, g1);
         ColGraph G1;
         collapse_graph (ncol, n1, g1, G, G1);
         n2 = pairwise_match (n1, G1.ptr, G1.nbr, G1.w, g2);
         for (int c = 0; c < ncol; c++) N.agg[c] = g2[g1[c]];
      }
      N.nagg = n2;
      Nat C;
      int64_t ncr = 0;
      const int split = K.split, pocket = K.pocket;
      const double theta = K.theta, tau = K.tau;
      if (geo && split) {
         // connectivity-aware coarse cells inside the geometric groups (see split_aggregate)
         auto t0 = clk::now ();
         SplitResult R;
         split_aggregate (N.L, N.blk_start, N.col_of, N.ktop, N.agg, cgi, cgj, cgt, N.gt, pocket, theta, tau, R);
         t_graph += secs (t0);
         ncr = R.blk_start.back ();
         if (ncr >= N.L.n) break;                       // no coarsening possible
         if (verbose)
            printf ("(%d) multilevel: level %d -> %d: %d columns in %d groups -> %d coarse columns (%d stubs), %d leaf stubs absorbed\n", rank, level0 + l, level0 + l + 1, ncol, n2,
                    (int) R.blk_start.size () - 1, R.stubs, R.absorbed);
         n2 = (int) R.blk_start.size () - 1;
         N.cmap.swap (R.cmap);
         C.blk_start.swap (R.blk_start);
         C.ktop.swap (R.ktop);
         C.gi.swap (R.gi); C.gj.swap (R.gj); C.gt.swap (R.gt);
      } else {
         if (n2 >= ncol) break;                         // no coarsening possible
         // coarse columns: length = longest member
         std::vector<int> clen (n2, 0);
         for (int c = 0; c < ncol; c++) clen[N.agg[c]] = std::max (clen[N.agg[c]], N.blk_start[c + 1] - N.blk_start[c]);
         C.gi.swap (cgi); C.gj.swap (cgj); C.gt.swap (cgt);
         C.blk_start.assign (n2 + 1, 0);
         C.ktop.assign (n2, 0);
         for (int a = 0; a < n2; a++) C.blk_start[a + 1] = C.blk_start[a] + clen[a];
         ncr = C.blk_start[n2];
         N.cmap.resize (N.L.n);
         for (int c = 0; c < ncol; c++)
            for (int r = N.blk_start[c]; r < N.blk_start[c + 1]; r++) N.cmap[r] = C.blk_start[N.agg[c]] + (r - N.blk_start[c]);
      }
      { auto t0 = clk::now (); galerkin (N.L, N.cmap, ncr, C.L); t_galerkin += secs (t0); }
      C.col_of.resize (ncr);
      for (int a = 0; a < n2; a++)
         for (int r = C.blk_start[a]; r < C.blk_start[a + 1]; r++) C.col_of[r] = a;
      nat.push_back (std::move (C));
   }

}

}  // namespace

// ================================================================ host-only plan (tests)
extern "C" int nkp_ml_plan_host (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const int32_t *blk_start, int64_t nblk,
                                 const int32_t *col_i, const int32_t *col_j, int coupled_tracer_cnt, int max_levels, int coarsest_rows, int64_t capacity,
                                 int *n_levels, int64_t *rows, int32_t *cmap, int32_t *col_of)
{
   if (n <= 0 || !rowptr || !colind || !val || !blk_start || nblk <= 0 || !n_levels || !rows || !cmap || !col_of) return NKP_EINVAL;
   if (max_levels <= 0) max_levels = 12;
   std::vector<Nat> nat (1);
   SetupTimes T;
   nkp_tuning tune;
   nkp_default_tuning (&tune);                     // test entry point: defaults + environment
   use_setup_knobs (tune);
   const PlanKnobs K = plan_knobs (tune);
   init_first_nat (nat[0], n, rowptr, colind, val, blk_start, nblk, col_i, col_j, nullptr, coupled_tracer_cnt, true, T);
   extend_nat_levels (nat, 0, (int) nblk, coupled_tracer_cnt, max_levels, coarsest_rows, 0, 0, K, T);
   if (tune.plan_times) printf ("nkp_ml_plan_host: %.2f s low-order twin, %.2f s graphs + aggregation, %.2f s Galerkin products\n", T.low, T.graph, T.galerkin);
   *n_levels = (int) nat.size ();
   int64_t qc = 0, qo = 0;
   for (size_t l = 0; l < nat.size (); l++) {
      rows[l] = nat[l].L.n;
      if (l + 1 < nat.size ()) {
         if (qc + nat[l].L.n > capacity || qo + nat[l + 1].L.n > capacity) return NKP_ENOMEM;
         std::copy (nat[l].cmap.begin (), nat[l].cmap.end (), cmap + qc);
         std::copy (nat[l + 1].col_of.begin (), nat[l + 1].col_of.end (), col_of + qo);
         qc += nat[l].L.n;
         qo += nat[l + 1].L.n;
      }
   }
   return 0;
}

// ================================================================ setup
namespace {

using setup_clk = std::chrono::steady_clock;
inline double secs_since (setup_clk::time_point a) { return std::chrono::duration<double> (setup_clk::now () - a).count (); }

struct DevTimes { double rb = 0.0, up = 0.0, fac = 0.0, lay = 0.0, map = 0.0, perm = 0.0, dev = 0.0, twin = 0.0, agg = 0.0, galerkin = 0.0; };

#define ML_FAIL(code, ...) do { snprintf (err, errlen, __VA_ARGS__); return (code); } while (0)

// Column blocks of a level whose colour-major operator is on the device (V.L with f64 values): half bandwidth, band LU of
// every column, lane layouts; then the f64 values are dropped if the cycle reads the f32 copy.  pblk = row offsets of the
// columns in colour-major order (host).
int finish_level_columns (MlHierarchy &H, MlLevel &V, int l, const std::vector<int> &pblk, int ncol, int ncol0, const int *h_prow, hipStream_t st,
                          char *err, size_t errlen, DevTimes &T)
{
   const int64_t nl = V.n;
   V.B.n = nl;
   V.B.nblk = ncol;
   V.B.tune = H.tune;
   if (!upload (&V.B.blk_start, pblk.data (), pblk.size (), &H.device_bytes)) ML_FAIL (-2, "multilevel setup: device allocation failed");
   int *dint = nullptr;
   size_t dummy = 0;
   std::vector<int> zeros (8, 0);
   auto t_fac0 = setup_clk::now ();
   if (!upload (&dint, zeros.data (), 8, &dummy)) ML_FAIL (-2, "multilevel setup: device allocation failed");
   launch_colblock_measure (V.L, V.B, dint, st);
   int meas[3];
   (void) hipMemcpyAsync (meas, dint, sizeof meas, hipMemcpyDeviceToHost, st);
   (void) hipStreamSynchronize (st);
   if (meas[1] > 0) { (void) hipFree (dint); ML_FAIL (-4, "multilevel setup: level %d has %d rows without a diagonal entry", l, meas[1]); }
   V.B.max_len = meas[2];
   V.B.P = meas[0] <= 1 ? 1 : meas[0] <= 2 ? 2 : 4;
   if (!upload (&V.B.fac, (const double *) nullptr, (size_t) (2 * V.B.P + 1) * (size_t) nl, &H.device_bytes)) { (void) hipFree (dint); ML_FAIL (-2, "multilevel setup: device allocation failed"); }
   (void) hipMemsetAsync (dint, 0, 8 * sizeof (int), st);
   launch_colblock_factor (V.L, V.B, dint, st);
   int st2[2];
   (void) hipMemcpyAsync (st2, dint, sizeof st2, hipMemcpyDeviceToHost, st);
   (void) hipStreamSynchronize (st);
   (void) hipFree (dint);
   if (st2[0] != 0) ML_FAIL (-4, "multilevel setup: zero pivot in a column block of level %d (row %d)", l, st2[0] - 1);
   V.B.dropped = st2[1];
   T.fac += secs_since (t_fac0);
   {
      auto t_lay0 = setup_clk::now ();
      const int ranges[3] = { 0, ncol0, ncol };
      const int lrc = colblock_build_lane_layout (V.B, pblk.data (), ranges, 2, V.color_grp, &H.device_bytes, st, H.f32, H.fused ? h_prow : nullptr);
      if (lrc != 0) ML_FAIL (-3, "multilevel setup: lane layout of level %d failed (HIP error %d)", l, lrc);
      V.wave_columns = ncol <= H.tune->col_wave_max && V.B.dropped == 0;
      // ... and their half sweeps are one launch each (residual of the column's rows + its band solve, gs_wave_kernel)
      V.wave_fused = V.wave_columns && (H.tune->ml_wave_fused == 1 || (H.tune->ml_wave_fused > 1 && ncol <= H.tune->ml_wave_fused));
      if (V.wave_fused) {
         if (!upload (&V.B.wave_desc, (const int *) nullptr, (size_t) 4 * (size_t) ncol, &H.device_bytes)) ML_FAIL (-2, "multilevel setup: device allocation failed");
         launch_build_wave_desc (V.L, V.B, st);
      }
      // the block-per-group fused kernel serves matching storage only (f32 operator with f32 factors, or f64 with f64)
      if (V.B.gs_ok && !((V.B.fac_tf && V.L.valf) || (V.B.fac_t && !V.L.valf))) V.B.gs_ok = 0;
      T.lay += secs_since (t_lay0);
   }
   // f32 storage mode: the f64 copy of the level operator was only needed to factor the column blocks
   if (V.L.valf && V.L.val) {
      (void) hipStreamSynchronize (st);
      (void) hipFree (V.L.val);
      V.L.val = nullptr;
      H.device_bytes -= ((size_t) V.L.nnz + 2) * sizeof (double);
   }
   return 0;
}

// one level from host arrays (natural order in N, next level in C or NULL): colour-major operator, uploads, column blocks,
// transfer maps, dense inverse of the last level
int finalize_host_level (MlHierarchy &H, int l, int nlev, Nat &N, Nat *Cn, int verbose, int rank, hipStream_t st, char *err, size_t errlen, DevTimes &T)
{
   MlLevel &V = H.lev[l];
   const int64_t nl = N.L.n;
   const int ncol = (int) N.blk_start.size () - 1;
   V.n = nl;
   // permuted CSR: row new = perm[new]; columns relabelled through inv, then sorted
   std::vector<int> prow (nl + 1, 0);
   auto t_perm0 = setup_clk::now ();
   RawInts pcol;                                    // sized without a fill: the row-parallel loop below writes every entry
   RawDoubles pval;
   pcol.resize (N.L.colind.size ());
   pval.resize (N.L.colind.size ());
   for (int64_t i = 0; i < nl; i++) prow[i + 1] = prow[i] + (N.L.rowptr[N.perm[i] + 1] - N.L.rowptr[N.perm[i]]);
   for_row_chunks (nl, [&] (int, int64_t i0, int64_t i1) {
      std::vector<std::pair<int, double>> tmp;
      for (int64_t i = i0; i < i1; i++) {
         const int o = N.perm[i];
         tmp.clear ();
         for (int e = N.L.rowptr[o]; e < N.L.rowptr[o + 1]; e++) tmp.emplace_back (N.inv[N.L.colind[e]], N.L.val[e]);
         std::sort (tmp.begin (), tmp.end ());
         int q = prow[i];
         for (auto &t : tmp) { pcol[q] = t.first; pval[q] = t.second; q++; }
      }
   });
   T.perm += secs_since (t_perm0);
   auto t_dev0 = setup_clk::now ();
   // permuted column blocks
   std::vector<int> pblk;
   pblk.reserve (ncol + 1);
   pblk.push_back (0);
   for (int pass = 0; pass < 2; pass++)
      for (int c = 0; c < ncol; c++)
         if (N.colour[c] == pass) pblk.push_back (pblk.back () + (N.blk_start[c + 1] - N.blk_start[c]));
   V.color_blk[0] = 0;
   V.color_blk[1] = N.ncol0;
   V.color_blk[2] = ncol;
   const int rows0 = pblk[N.ncol0];
   V.rows0 = rows0;
   // row blocks per colour (must not straddle the colour boundary)
   int *rb0 = nullptr, *rb1 = nullptr, nrb0 = 0, nrb1 = 0;
   auto t_rb0 = setup_clk::now ();
   build_rowblocks_host (rows0, prow.data (), &rb0, &nrb0);
   {
      std::vector<int> shifted (nl - rows0 + 1);
      for (int64_t i = rows0; i <= nl; i++) shifted[i - rows0] = prow[i] - prow[rows0];
      build_rowblocks_host (nl - rows0, shifted.data (), &rb1, &nrb1);
   }
   std::vector<int> rb (nrb0 + nrb1 + 1);
   for (int i = 0; i <= nrb0; i++) rb[i] = rb0[i];
   for (int i = 1; i <= nrb1; i++) rb[nrb0 + i] = rows0 + rb1[i];
   if (nl - rows0 == 0) nrb1 = 0;
   if (rows0 == 0) { nrb0 = 0; }
   free (rb0);
   free (rb1);
   V.color_rb[0] = 0;
   V.color_rb[1] = nrb0;
   V.color_rb[2] = nrb0 + nrb1;
   V.L.n = nl;
   V.L.nnz = prow[nl];
   V.L.nrowblk = nrb0 + nrb1;
   V.L.tune = H.tune;
   T.rb += secs_since (t_rb0);
   auto t_up0 = setup_clk::now ();
   bool ok = upload (&V.L.rowptr, prow.data (), (size_t) nl + 1, &H.device_bytes) &&
             upload_padded (&V.L.colind, pcol.data (), (size_t) prow[nl], 2, &H.device_bytes) &&
             upload_padded (&V.L.val, pval.data (), (size_t) prow[nl], 2, &H.device_bytes) &&
             upload (&V.L.rowblk, rb.data (), rb.size (), &H.device_bytes) &&
             upload (&V.x, (const double *) nullptr, (size_t) nl, &H.device_bytes) &&
             upload (&V.x2, (const double *) nullptr, (size_t) nl, &H.device_bytes) &&
             upload (&V.b, (const double *) nullptr, (size_t) nl, &H.device_bytes) &&
             upload (&V.r, (const double *) nullptr, (size_t) nl, &H.device_bytes);
   if (ok) ok = attach_spmv_codes (V.L, prow.data (), pcol.data (), rb.data (), &H.device_bytes) == 0;
   if (ok && H.f32 && l < nlev - 1) {
      std::vector<float, RawAlloc<float>> vf;
      vf.resize ((size_t) prow[nl]);
      for_row_chunks (nl, [&] (int, int64_t i0, int64_t i1) {
         for (int64_t e = prow[i0]; e < prow[i1]; e++) vf[(size_t) e] = (float) pval[(size_t) e];
      });
      ok = upload_padded (&V.L.valf, vf.data (), vf.size (), 2, &H.device_bytes);
   }
   T.up += secs_since (t_up0);
   if (!ok) ML_FAIL (-2, "multilevel setup: device allocation failed at level %d", l);
   if (l == 0 && !upload (&H.perm0, N.perm.data (), (size_t) nl, &H.device_bytes)) ML_FAIL (-2, "multilevel setup: device allocation failed");

   const int dense_max = H.tune->ml_dense_max;
   // the last level is solved with a dense inverse when it is small enough; otherwise (rough bathymetry can leave
   // thousands of pocket stubs that nothing absorbs) it is relaxed like the others, with many sweeps
   const bool dense_last = (l == nlev - 1) && nl <= dense_max;
   if (!dense_last) {
      const int frc = finish_level_columns (H, V, l, pblk, ncol, N.ncol0, prow.data (), st, err, errlen, T);
      if (frc) return frc;
   }
   if (l < nlev - 1) {
      // transfer operators in permuted orders
      auto t_map0 = setup_clk::now ();
      Nat &C = *Cn;
      const int64_t nc = C.L.n;
      V.nc = nc;
      std::vector<int> cmap_p (nl);
      for (int64_t i = 0; i < nl; i++) cmap_p[i] = C.inv[N.cmap[N.perm[i]]];
      std::vector<int> rptr (nc + 1, 0), ridx (nl);
      for (int64_t i = 0; i < nl; i++) rptr[cmap_p[i] + 1]++;
      for (int64_t I = 0; I < nc; I++) rptr[I + 1] += rptr[I];
      {
         std::vector<int> fill (rptr.begin (), rptr.end () - 1);
         for (int64_t i = 0; i < nl; i++) ridx[fill[cmap_p[i]]++] = (int) i;
      }
      if (!(upload (&V.cmap, cmap_p.data (), (size_t) nl, &H.device_bytes) && upload (&V.rptr, rptr.data (), (size_t) nc + 1, &H.device_bytes) &&
            upload (&V.ridx, ridx.data (), (size_t) nl, &H.device_bytes)))
         ML_FAIL (-2, "multilevel setup: device allocation failed");
      T.map += secs_since (t_map0);
   }
   if (dense_last) {
      // coarsest level: dense inverse (permuted order)
      std::vector<double> dense;
      auto fill_dense = [&] () {
         dense.assign ((size_t) nl * nl, 0.0);
         for (int64_t i = 0; i < nl; i++)
            for (int e = prow[i]; e < prow[i + 1]; e++) dense[(size_t) i * nl + pcol[e]] = pval[e];
      };
      if (H.tune->ml_host_inverse) {
         fill_dense ();
         if (!dense_inverse ((int) nl, dense)) ML_FAIL (-4, "multilevel setup: coarsest operator is singular");
         if (!upload (&H.coarse_inv, dense.data (), dense.size (), &H.device_bytes)) ML_FAIL (-2, "multilevel setup: device allocation failed");
      } else {
         // blocked elimination on the matrix cores (dense.hip); a pivot it does not trust sends the level to the pivoted routine
         auto t_inv0 = setup_clk::now ();
         const int brc = dense_inverse_blocked_device ((int) nl, prow.data (), pcol.data (), pval.data (), &H.coarse_inv, H.f32 ? &H.coarse_invf : nullptr,
                                                       &H.coarse_ldf, &H.device_bytes, st);
         if (brc < 0) ML_FAIL (-2, "multilevel setup: device allocation failed (dense inverse of %lld rows)", (long long) nl);
         if (verbose) printf ("(%d) multilevel: dense inverse of %lld rows: %s, %.3f s\n", rank, (long long) nl, brc == 0 ? "blocked elimination" : "a pivot too small for it, pivoted routine instead", secs_since (t_inv0));
         if (brc > 0) fill_dense ();
         if (brc > 0 && !dense_inverse_device ((int) nl, dense, &H.coarse_inv, &H.device_bytes, st))
            ML_FAIL (-4, "multilevel setup: coarsest operator is singular (or the device is out of memory)");
      }
   }
   if (verbose)
      printf ("(%d) multilevel: level %d: %lld rows, %lld entries, %d columns (%d + %d by colour)%s\n", rank, l, (long long) nl,
              (long long) prow[nl], ncol, N.ncol0, ncol - N.ncol0, l < nlev - 1 ? "" : dense_last ? ", dense solve" : ", relaxed (too large for a dense inverse)");
   T.dev += secs_since (t_dev0);
   return 0;
}

// ---- levels whose operator is built on the device (mlsetup.hip)
struct DevLevel {
   mls::DevCsr L;                                                     // natural order
   int *blk_start = nullptr, *col_of = nullptr, *ktop = nullptr;      // per column / per row / per column
   int *perm = nullptr, *inv = nullptr;                               // colour-major maps
   int *cmap = nullptr;                                               // natural row -> natural row of the next level
   std::vector<int> newstart, pblk;                                   // host: colour-major start row of every column, permuted blocks
   void free_all ()
   {
      L.free_all ();
      for (int **p : { &blk_start, &col_of, &ktop, &perm, &inv, &cmap })
         if (*p) { (void) hipFree (*p); *p = nullptr; }
   }
};

// column arrays of a level on the device + its colour-major maps; N holds the host column arrays (blk_start, ktop, gi, gj)
int dev_level_columns (DevLevel &D, Nat &N, int64_t n, hipStream_t st)
{
   const int ncol = (int) N.blk_start.size () - 1;
   size_t dummy = 0;
   N.colour.resize (ncol);
   for (int c = 0; c < ncol; c++) N.colour[c] = (N.gi[c] + N.gj[c]) & 1;
   colour_major_columns (N, N.colour, D.newstart, D.pblk);
   int *d_newstart = nullptr;
   bool ok = upload (&D.blk_start, N.blk_start.data (), (size_t) ncol + 1, &dummy) && upload (&D.ktop, N.ktop.data (), (size_t) ncol, &dummy) &&
             upload (&D.col_of, (const int *) nullptr, (size_t) n, &dummy) && upload (&D.perm, (const int *) nullptr, (size_t) n, &dummy) &&
             upload (&D.inv, (const int *) nullptr, (size_t) n, &dummy) && upload (&d_newstart, D.newstart.data (), (size_t) ncol, &dummy);
   if (!ok) { if (d_newstart) (void) hipFree (d_newstart); return 1; }
   mls::rows_to_cols (D.blk_start, ncol, D.col_of, st);
   mls::colour_major_maps (D.blk_start, D.col_of, d_newstart, n, D.perm, D.inv, st);
   const hipError_t e = hipStreamSynchronize (st);
   (void) hipFree (d_newstart);
   return e == hipSuccess ? 0 : 1;
}

// a level that has a coarser one, from its device-resident natural operator: colour-major operator, row blocks, f32 copy,
// column blocks, transfer maps.  inv_next = colour-major map of the next level (device).
int finalize_device_level (MlHierarchy &H, int l, DevLevel &D, Nat &N, const int *inv_next, int64_t nc, int verbose, int rank, hipStream_t st,
                           char *err, size_t errlen, DevTimes &T)
{
   MlLevel &V = H.lev[l];
   const int64_t nl = D.L.n;
   const int ncol = (int) N.blk_start.size () - 1;
   V.n = nl;
   auto t0 = setup_clk::now ();
   int *prow = nullptr, *pcol = nullptr;
   double *pval = nullptr;
   int rc = mls::permute_operator (D.L, D.perm, D.inv, &prow, &pcol, &pval, 2, st);
   if (rc) ML_FAIL (-2, "multilevel setup: colour-major operator of level %d failed on the device (HIP error %d)", l, rc);
   V.L.n = nl;
   V.L.nnz = D.L.nnz;
   V.L.tune = H.tune;
   V.L.rowptr = prow;
   V.L.colind = pcol;
   V.L.val = pval;
   H.device_bytes += ((size_t) nl + 1) * sizeof (int) + ((size_t) D.L.nnz + 2) * (sizeof (int) + sizeof (double));
   T.perm += secs_since (t0);
   t0 = setup_clk::now ();
   V.color_blk[0] = 0;
   V.color_blk[1] = N.ncol0;
   V.color_blk[2] = ncol;
   const int rows0 = D.pblk[N.ncol0];
   V.rows0 = rows0;
   {
      // row blocks per colour (must not straddle the colour boundary)
      int *rb0 = nullptr, *rb1 = nullptr, nrb0 = 0, nrb1 = 0;
      if ((rc = mls::row_blocks (prow, 0, rows0, &rb0, &nrb0, st)) || (rc = mls::row_blocks (prow, rows0, nl, &rb1, &nrb1, st))) {
         if (rb0) (void) hipFree (rb0);
         ML_FAIL (-2, "multilevel setup: row blocks of level %d failed on the device (HIP error %d)", l, rc);
      }
      int *rb = nullptr;
      bool ok = hipMalloc ((void **) &rb, (size_t) (nrb0 + nrb1 + 1) * sizeof (int)) == hipSuccess;
      if (ok && nrb0) ok = hipMemcpyAsync (rb, rb0, (size_t) nrb0 * sizeof (int), hipMemcpyDeviceToDevice, st) == hipSuccess;
      if (ok && nrb1) ok = hipMemcpyAsync (rb + nrb0, rb1, (size_t) (nrb1 + 1) * sizeof (int), hipMemcpyDeviceToDevice, st) == hipSuccess;
      if (ok && !nrb1) { const int last = (int) nl; ok = hipMemcpyAsync (rb + nrb0, &last, sizeof (int), hipMemcpyHostToDevice, st) == hipSuccess; }
      if (ok) ok = hipStreamSynchronize (st) == hipSuccess;
      if (rb0) (void) hipFree (rb0);
      if (rb1) (void) hipFree (rb1);
      if (!ok) { if (rb) (void) hipFree (rb); ML_FAIL (-2, "multilevel setup: device allocation failed at level %d", l); }
      V.L.rowblk = rb;
      V.L.nrowblk = nrb0 + nrb1;
      V.color_rb[0] = 0;
      V.color_rb[1] = nrb0;
      V.color_rb[2] = nrb0 + nrb1;
      H.device_bytes += (size_t) (nrb0 + nrb1 + 1) * sizeof (int);
   }
   T.rb += secs_since (t0);
   t0 = setup_clk::now ();
   bool ok = upload (&V.x, (const double *) nullptr, (size_t) nl, &H.device_bytes) && upload (&V.x2, (const double *) nullptr, (size_t) nl, &H.device_bytes) &&
             upload (&V.b, (const double *) nullptr, (size_t) nl, &H.device_bytes) && upload (&V.r, (const double *) nullptr, (size_t) nl, &H.device_bytes);
   if (ok && H.f32) {
      ok = upload (&V.L.valf, (const float *) nullptr, (size_t) D.L.nnz + 2, &H.device_bytes) &&
           hipMemsetAsync (V.L.valf + D.L.nnz, 0, 2 * sizeof (float), st) == hipSuccess;
      if (ok) mls::to_float (pval, V.L.valf, D.L.nnz, st);
   }
   if (!ok) ML_FAIL (-2, "multilevel setup: device allocation failed at level %d", l);
   if (l == 0) {
      // level-0 row i holds original row perm0[i]
      if (!upload (&H.perm0, (const int *) nullptr, (size_t) nl, &H.device_bytes) ||
          hipMemcpyAsync (H.perm0, D.perm, (size_t) nl * sizeof (int), hipMemcpyDeviceToDevice, st) != hipSuccess)
         ML_FAIL (-2, "multilevel setup: device allocation failed");
   }
   T.up += secs_since (t0);
   const int frc = finish_level_columns (H, V, l, D.pblk, ncol, N.ncol0, nullptr, st, err, errlen, T);
   if (frc) return frc;
   {
      // transfer operators in colour-major orders
      t0 = setup_clk::now ();
      V.nc = nc;
      if (!upload (&V.cmap, (const int *) nullptr, (size_t) nl, &H.device_bytes)) ML_FAIL (-2, "multilevel setup: device allocation failed");
      mls::permuted_cmap (D.cmap, D.perm, inv_next, nl, V.cmap, st);
      if ((rc = mls::inverse_map (V.cmap, nl, nc, &V.rptr, &V.ridx, st))) ML_FAIL (-2, "multilevel setup: transfer maps of level %d failed on the device (HIP error %d)", l, rc);
      H.device_bytes += ((size_t) nc + 1 + (size_t) nl) * sizeof (int);
      T.map += secs_since (t0);
   }
   if (verbose)
      printf ("(%d) multilevel: level %d: %lld rows, %lld entries, %d columns (%d + %d by colour), built on the device\n", rank, l, (long long) nl,
              (long long) D.L.nnz, ncol, N.ncol0, ncol - N.ncol0);
   return 0;
}

}  // namespace

int ml_setup (MlHierarchy &H, int64_t n, const int *rowptr, const int *colind, const double *val,
              const int *blk_start_in, int64_t nblk, const int *col_i, const int *col_j, const int *col_t, int tracer_cnt, int max_levels, int nu, int coarsest_rows, int verbose, int rank,
              hipStream_t st, char *err, size_t errlen, const nkp_tuning &tune, const CsrDev *A_dev)
{
   DevTimes T;
   SetupTimes TH;
   H.tune = &tune;                                // the caller's (solver's) copy outlives the hierarchy
   use_setup_knobs (tune);
   H.nu = nu < 1 ? 1 : nu;
   H.f32 = tune.ml_f32 != 0;                      // level operators and factors stored in f32, arithmetic in f64
   // one launch per half sweep (gs_fused_kernel): bit-identical, but measured slower than the two tuned kernels -- 1 degree
   // cycle 3.62 against 2.65 ms, 3 degree 1.56 against 1.26 ms: a workgroup walks its group's 2-3 row blocks one after the
   // other, each exposing the stream -> gather -> row-sum latency chain that the standalone SpMV hides with one block per
   // workgroup and hundreds of workgroups in flight -- so it stays off (ml_fused = 1 turns it on)
   H.fused = tune.ml_fused != 0;
   H.nu_coarse = tune.ml_smooth_coarse >= 1 ? tune.ml_smooth_coarse : H.nu;
   if (tune.ml_coarse_from >= 1) H.coarse_from = tune.ml_coarse_from;
   // 1.1: the Galerkin operators of piecewise-constant cells are too stiff where lateral mixing matters, so a slightly
   // over-weighted coarse correction helps (1 degree: 0.9 -> 77 iterations, 1.0 -> 69, 1.1 -> 64, 1.2 -> 68, 1.35 -> 95)
   H.omega = tune.ml_omega > 0.0 ? tune.ml_omega : 1.1;
   H.gamma_from = tune.ml_gamma_from;
   H.gamma_to = tune.ml_gamma_to;
   if (max_levels <= 0) max_levels = 12;
   const PlanKnobs K = plan_knobs (tune);
   const auto t_begin = setup_clk::now ();

   // Levels with at least dev_min rows are built by the kernels of mlsetup.hip, the rest by the host routines above (a
   // level of a few 10^4 rows costs less on the host than the launches and round trips of the device passes); both build
   // the same hierarchy entry for entry.  The device passes cover the default construction only: geometric groups with
   // connectivity-aware cells, no edge threshold, no 2-byte column codes, no fused half sweeps.
   const int64_t dev_min = tune.ml_device_min;
   const bool device_ok = col_i && col_j && K.split && K.theta == 0.0 && !tune.spmv_compress && !H.fused && dev_min >= 0;

   std::vector<Nat> hnat;              // host-built levels (the first of them may have been handed over by the device path)
   int l0 = 0;                         // index of hnat[0] in the hierarchy
   int ndev_levels = 0;
   if (device_ok && n >= dev_min && !((1 >= max_levels) || (n <= coarsest_rows) || nblk <= 4)) {
      // ---------------- device path
      DevLevel D;
      Nat N;
      init_first_nat (N, n, rowptr, colind, val, blk_start_in, nblk, col_i, col_j, col_t, tracer_cnt, false, TH);
      N.L.n = n;
      size_t dummy = 0;
      {
         // the matrix: the caller's device copy when there is one, else a temporary upload
         auto t0 = setup_clk::now ();
         int *a_row = nullptr, *a_col = nullptr;
         double *a_val = nullptr;
         const int nnz = rowptr[n];
         if (!A_dev) {
            if (!(upload (&a_row, rowptr, (size_t) n + 1, &dummy) && upload (&a_col, colind, (size_t) nnz, &dummy) && upload (&a_val, val, (size_t) nnz, &dummy))) {
               for (void *p : { (void *) a_row, (void *) a_col, (void *) a_val }) if (p) (void) hipFree (p);
               ML_FAIL (-2, "multilevel setup: device allocation failed");
            }
         }
         int rc = dev_level_columns (D, N, n, st);
         if (!rc) rc = mls::twin (n, A_dev ? A_dev->rowptr : a_row, A_dev ? A_dev->colind : a_col, A_dev ? A_dev->val : a_val, D.col_of, D.L, st);
         for (void *p : { (void *) a_row, (void *) a_col, (void *) a_val }) if (p) (void) hipFree (p);
         if (rc) { D.free_all (); ML_FAIL (-2, "multilevel setup: low-order twin failed on the device (HIP error %d)", rc); }
         T.twin += secs_since (t0);
      }
      for (int l = 0;; l++) {
         // D / N = level l, resident on the device, with a coarser level to come unless the aggregation stalls
         const int ncol = (int) N.blk_start.size () - 1;
         std::vector<int> cgi, cgj, cgt;
         const int ng = geo_groups (N, group_shift (K, l, (int) nblk, tracer_cnt), N.agg, cgi, cgj, cgt);
         N.nagg = ng;
         auto t0 = setup_clk::now ();
         mls::AggregateIn ain;
         ain.n = D.L.n; ain.ncol = ncol;
         ain.rowptr = D.L.rowptr; ain.colind = D.L.colind; ain.val = D.L.val;
         ain.blk_start = D.blk_start; ain.col_of = D.col_of; ain.ktop = D.ktop;
         ain.h_blk_start = N.blk_start.data (); ain.h_ktop = N.ktop.data (); ain.h_group = N.agg.data (); ain.h_col_t = N.gt.data ();
         ain.pocket = K.pocket; ain.tau = K.tau;
         mls::AggregateOut aout;
         int rc = mls::aggregate (ain, aout, st);
         if (rc) { D.free_all (); ML_FAIL (-2, "multilevel setup: aggregation of level %d failed on the device (HIP error %d)", l, rc); }
         T.agg += secs_since (t0);
         D.cmap = aout.cmap;
         const int64_t ncr = aout.blk_start.back ();
         Nat C;
         DevLevel DC;
         bool stalled = ncr >= D.L.n;                  // no coarsening possible: level l is the last one
         if (!stalled) {
            if (verbose)
               printf ("(%d) multilevel: level %d -> %d: %d columns in %d groups -> %d coarse columns (%d stubs), %d leaf stubs absorbed\n", rank, l, l + 1, ncol, ng,
                       (int) aout.blk_start.size () - 1, aout.stubs, aout.absorbed);
            const int ncc = (int) aout.blk_start.size () - 1;
            C.blk_start.swap (aout.blk_start);
            C.ktop.swap (aout.ktop);
            C.gi.resize (ncc); C.gj.resize (ncc); C.gt.resize (ncc);
            for (int q = 0; q < ncc; q++) { const int g = aout.group[q]; C.gi[q] = cgi[g]; C.gj[q] = cgj[g]; C.gt[q] = cgt[g]; }
            t0 = setup_clk::now ();
            rc = mls::galerkin (D.L, D.cmap, ncr, DC.L, st);
            if (!rc) rc = dev_level_columns (DC, C, ncr, st);
            if (rc) { D.free_all (); DC.free_all (); ML_FAIL (-2, "multilevel setup: Galerkin product of level %d failed on the device (HIP error %d)", l, rc); }
            C.L.n = ncr;
            T.galerkin += secs_since (t0);
         }
         // what becomes of the next level (or of this one, if it is the last): handed to the host routines when it is small
         // or final
         const bool next_last = stalled || (l + 2 >= max_levels) || (ncr <= coarsest_rows) || (int) C.blk_start.size () - 1 <= 4;
         const bool hand_over = stalled || next_last || ncr < dev_min;
         if (!stalled) {
            H.lev.resize ((size_t) l + 1);
            const int frc = finalize_device_level (H, l, D, N, DC.inv, ncr, verbose, rank, st, err, errlen, T);
            if (frc) { D.free_all (); DC.free_all (); return frc; }
            ndev_levels = l + 1;
         }
         if (hand_over) {
            // download the natural operator of the level the host continues from
            DevLevel &S = stalled ? D : DC;
            Nat &M = stalled ? N : C;
            M.L.n = S.L.n;
            M.L.rowptr.resize ((size_t) S.L.n + 1);
            M.L.colind.resize ((size_t) S.L.nnz);
            M.L.val.resize ((size_t) S.L.nnz);
            bool ok = hipMemcpy (M.L.rowptr.data (), S.L.rowptr, ((size_t) S.L.n + 1) * sizeof (int), hipMemcpyDeviceToHost) == hipSuccess;
            if (ok && S.L.nnz) ok = hipMemcpy (M.L.colind.data (), S.L.colind, (size_t) S.L.nnz * sizeof (int), hipMemcpyDeviceToHost) == hipSuccess &&
                                    hipMemcpy (M.L.val.data (), S.L.val, (size_t) S.L.nnz * sizeof (double), hipMemcpyDeviceToHost) == hipSuccess;
            const int ncm = (int) M.blk_start.size () - 1;
            M.col_of.resize ((size_t) S.L.n);
            for (int c = 0; c < ncm; c++)
               for (int r = M.blk_start[c]; r < M.blk_start[c + 1]; r++) M.col_of[r] = c;
            M.colour.clear (); M.agg.clear ();
            l0 = stalled ? l : l + 1;
            hnat.clear ();
            hnat.push_back (std::move (M));
            D.free_all ();
            DC.free_all ();
            if (!ok) ML_FAIL (-3, "multilevel setup: download of level %d failed", l0);
            if (!stalled && !next_last) extend_nat_levels (hnat, l0, (int) nblk, tracer_cnt, max_levels, coarsest_rows, verbose, rank, K, TH);
            else {
               // a final level: only its colour-major order is missing
               std::vector<Nat> one;
               one.push_back (std::move (hnat[0]));
               extend_nat_levels (one, l0, (int) nblk, tracer_cnt, l0 + 1, coarsest_rows, verbose, rank, K, TH);
               hnat.swap (one);
            }
            break;
         }
         D.free_all ();
         D = DC;                    // plain struct of pointers + two vectors
         DC = DevLevel ();
         N = std::move (C);
      }
   } else {
      hnat.resize (1);
      init_first_nat (hnat[0], n, rowptr, colind, val, blk_start_in, nblk, col_i, col_j, col_t, tracer_cnt, true, TH);
      extend_nat_levels (hnat, 0, (int) nblk, tracer_cnt, max_levels, coarsest_rows, verbose, rank, K, TH);
   }
   const double t_plan = secs_since (t_begin);

   // ---- host-built levels in colour-major order
   const int nlev = l0 + (int) hnat.size ();
   H.lev.resize (nlev);
   for (int l = l0; l < nlev; l++) {
      const int frc = finalize_host_level (H, l, nlev, hnat[(size_t) (l - l0)], l + 1 < nlev ? &hnat[(size_t) (l - l0 + 1)] : nullptr, verbose, rank, st, err, errlen, T);
      if (frc) return frc;
   }
   {
      // the small end of the cycle in one single-workgroup launch: the last levels whose rows add up to <= NKP_ML_TAIL_ROWS
      const int64_t cap = tune.ml_tail_rows;   // 0 = off: measured 1.7-5x SLOWER (1 degree cycle 4.57 against 2.68 ms) -- one workgroup is latency-bound on a single CU
      H.tail_from = -1;
      int64_t rows = 0;
      for (int l = nlev - 1; l >= 0 && nlev - l <= 8; l--) {
         rows += H.lev[l].n;
         if (rows > cap) break;
         if (l < nlev - 1) H.tail_from = l;           // at least two levels, else there is nothing to merge
      }
      if (verbose && H.tail_from >= 0) printf ("(%d) multilevel: levels %d..%d run as one single-workgroup launch\n", rank, H.tail_from, nlev - 1);
   }
   H.setup_seconds = secs_since (t_begin);
   H.levels_on_device = ndev_levels;
   if (verbose) {
      printf ("(%d) multilevel setup: %d of %d levels built on the device (%.3f s twin, %.3f s coarse cells, %.3f s Galerkin products); host levels: %.3f s twin, %.3f s "
              "column graphs + coarse cells, %.3f s Galerkin products; colour-major operators %.3f s, row blocks %.3f, uploads + f32 copies %.3f, column factors %.3f, "
              "lane layouts %.3f, transfer maps %.3f; hierarchy construction as a whole %.3f s, everything %.3f s\n",
              rank, ndev_levels, nlev, T.twin, T.agg, T.galerkin, TH.low, TH.graph, TH.galerkin, T.perm, T.rb, T.up, T.fac, T.lay, T.map, t_plan, H.setup_seconds);
      fflush (stdout);
   }
   return 0;
}
#undef ML_FAIL

void ml_free (MlHierarchy &H)
{
   for (MlLevel &V : H.lev) {
      void *ptrs[] = { V.L.rowptr, V.L.colind, V.L.val, V.L.valf, V.B.fac_tf, V.L.rowblk, V.L.codes, V.L.dict, V.L.dict_ptr, V.B.blk_start, V.B.fac, V.B.grp_b0, V.B.grp_nb, V.B.grp_maxlen, V.B.grp_base, V.B.grp_row0, V.B.col_slot, V.B.fac_t, V.B.gs_rb_ptr, V.B.gs_rb, V.B.wave_desc, V.cmap, V.rptr, V.ridx, V.x, V.x2, V.b, V.r, V.bx, V.bx2, V.bb, V.br };
      for (void *p : ptrs)
         if (p) (void) hipFree (p);
   }
   H.lev.clear ();
   if (H.perm0) (void) hipFree (H.perm0);
   if (H.coarse_inv) (void) hipFree (H.coarse_inv);
   if (H.coarse_invf) (void) hipFree (H.coarse_invf);
   H.coarse_invf = nullptr;
   H.perm0 = nullptr;
   H.coarse_inv = nullptr;
}

// ================================================================ cycle
// one Gauss-Seidel half sweep over colour c.  Fused path: one launch, x ping-pongs between the level's two buffers (the
// new values of colour c go where the other colour's current values are if the level is incoherent, else to the other
// buffer).  Two-kernel path: residual SpMV of the colour's rows, then the column solves accumulate into x in place.
// column solves of colour c: levels with few columns run one column per WAVE (colblock_apply_kernel: one round trip for the
// column's right-hand side and factors, the substitution by lane broadcasts) -- with thousands of idle wave slots its
// ~5 us beat the 12-16 us latency floor of the lane-per-column kernels, which only win when the chip is full
static void column_solves (const MlHierarchy &H, MlLevel &V, int c, const double *rhs, double *x, int accumulate, hipStream_t st)
{
   if (V.wave_columns) {
      if (H.f32) launch_colblock_apply_range_r32 (V.B, V.color_blk[c], V.color_blk[c + 1], rhs, x, accumulate, st);
      else launch_colblock_apply_range (V.B, V.color_blk[c], V.color_blk[c + 1], rhs, x, accumulate, st);
   } else
      launch_colblock_apply_lanes (V.B, V.color_grp[c], V.color_grp[c + 1], rhs, x, accumulate, st);
}

static void gs_half (const MlHierarchy &H, MlLevel &V, int c, bool fused, hipStream_t st)
{
   if (fused) {
      const int out = (V.cur[0] != V.cur[1]) ? V.cur[1 - c] : 1 - V.cur[c];
      const int rows0 = (int) V.rows0;
      // (the launchers cannot refuse: ml_setup clears gs_ok / wave_fused for a level whose storage they do not serve)
      if (V.wave_fused) launch_gs_wave (V.L, V.B, V.color_blk[c], V.color_blk[c + 1], V.xbuf (V.cur[0]), V.xbuf (V.cur[1]), rows0, V.b, V.xbuf (out), H.f32, st);
      else (void) launch_gs_fused (V.L, V.B, V.color_grp[c], V.color_grp[c + 1], V.xbuf (V.cur[0]), V.xbuf (V.cur[1]), rows0, V.b, V.xbuf (out), st);
      V.cur[c] = out;
      return;
   }
   launch_csr_residual_range (V.L, V.color_rb[c], V.color_rb[c + 1], V.x, V.b, V.r, st);
   column_solves (H, V, c, V.r, V.x, 1, st);
}

static void gs_sweep (const MlHierarchy &H, MlLevel &V, bool reverse, bool fused, hipStream_t st)
{
   for (int step = 0; step < 2; step++) gs_half (H, V, reverse ? 1 - step : step, fused, st);
}

static void ml_cycle (MlHierarchy &H, int l, hipStream_t st)
{
   MlLevel &V = H.lev[l];
   V.cur[0] = V.cur[1] = 0;
   if (l == H.tail_from && H.coarse_inv && H.gamma_to <= H.gamma_from && ml_tail_launch (H, l, st) == 0) return;
   if (l == (int) H.lev.size () - 1) {
      if (H.coarse_inv) {
         if (H.coarse_invf) launch_dense_matvec_f32 (H.coarse_invf, H.coarse_ldf, V.b, V.x, (int) V.n, st);
         else launch_dense_matvec (H.coarse_inv, V.b, V.x, (int) V.n, st);
         return;
      }
      // no dense inverse: many sweeps of the column smoother from x = 0 (what is left here is diagonally dominant)
      const int sweeps = H.tune->ml_coarsest_sweeps > 0 ? H.tune->ml_coarsest_sweeps : 30;
      launch_fill (V.x, 0.0, V.n, st);
      column_solves (H, V, 0, V.b, V.x, 0, st);
      gs_half (H, V, 1, false, st);
      for (int s = 1; s < sweeps; s++) gs_sweep (H, V, s & 1, false, st);
      return;
   }
   // ml_fused_max_cols: the fused half sweep only on levels with at most that many columns (the launch-bound end)
   const int fused_max = H.tune->ml_fused_max_cols;
   const bool fused = V.wave_fused || (H.fused && V.B.gs_ok && (fused_max <= 0 || V.color_grp[2] * V.B.gw <= fused_max));
   // pre-smoothing from x = 0: the first half-sweep needs no SpMV (r = b on colour 0)
   launch_fill (V.x, 0.0, V.n, st);
   if (fused) {
      // colour 0's first values go to the second buffer: the level starts incoherent, and the odd number of fused half
      // sweeps that follows (colour 1, then nu - 1 full sweeps) ends coherent
      if (V.wave_fused) column_solves (H, V, 0, V.b, V.x2, 0, st);
      else launch_colblock_apply_lanes (V.B, V.color_grp[0], V.color_grp[1], V.b, V.x2, 0, st);
      V.cur[0] = 1;
      gs_half (H, V, 1, true, st);
   } else {
      column_solves (H, V, 0, V.b, V.x, 0, st);
      gs_half (H, V, 1, false, st);
   }
   const int nu = (l >= H.coarse_from) ? H.nu_coarse : H.nu;
   for (int s = 1; s < nu; s++) gs_sweep (H, V, false, fused, st);
   // coarse-grid correction; levels in [gamma_from, gamma_to) repeat it on the updated residual, which by the
   // Galerkin property is the second coarse iteration of a W-cycle (NKP_ML_GAMMA_FROM / NKP_ML_GAMMA_TO, default off)
   MlLevel &C = H.lev[l + 1];
   const int gamma = (l >= H.gamma_from && l < H.gamma_to) ? 2 : 1;
   for (int g = 0; g < gamma; g++) {
      launch_csr_spmv (V.L, V.xnow (), V.r, V.b, 1, st);
      launch_restrict_sum (V.rptr, V.ridx, V.r, C.b, V.nc, st);
      ml_cycle (H, l + 1, st);
      launch_prolong_add (V.cmap, C.xnow (), V.xnow (), V.n, H.omega, st);
   }
   for (int s = 0; s < nu; s++) gs_sweep (H, V, true, fused, st);
}

void ml_apply (MlHierarchy &H, const double *r, double *z, hipStream_t st)
{
   MlLevel &V = H.lev[0];
   launch_gather (H.perm0, r, V.b, V.n, st);
   ml_cycle (H, 0, st);
   launch_scatter (H.perm0, V.xnow (), z, V.n, st);
}

// ================================================================ the cycle on K interleaved right-hand sides
int ml_batch_prepare (MlHierarchy &H, int K)
{
   if (K != 2 && K != 4 && K != 8) return -1;
   if (H.batch_K >= K) return 0;
   for (MlLevel &V : H.lev) {
      for (double **p : { &V.bx, &V.bx2, &V.bb, &V.br }) {
         if (*p) { (void) hipFree (*p); *p = nullptr; }
         const size_t bytes = (size_t) (V.n ? V.n : 1) * (size_t) K * sizeof (double);
         if (hipMalloc ((void **) p, bytes) != hipSuccess) return -2;
         if (hipMemset (*p, 0, bytes) != hipSuccess) return -2;
         H.device_bytes += bytes;
      }
   }
   H.batch_K = K;
   return 0;
}

static void column_solves_batch (const MlHierarchy &H, MlLevel &V, int K, int c, const double *rhs, double *x, int accumulate, hipStream_t st)
{
   // the packed lane layout has a two-system kernel; every other level takes the wave-per-column kernel (which reads the
   // f64 factors and rounds them like the f32 layouts store them: same values)
   if (V.wave_columns || launch_colblock_apply_lanes_batch (K, V.B, V.color_grp[c], V.color_grp[c + 1], rhs, x, accumulate, st) != 0)
      launch_colblock_apply_wave_batch (K, V.B, V.color_blk[c], V.color_blk[c + 1], rhs, x, accumulate, H.f32, st);
}

static void gs_half_batch (const MlHierarchy &H, MlLevel &V, int K, int c, hipStream_t st)
{
   if (V.wave_fused) {
      const int out = (V.bcur[0] != V.bcur[1]) ? V.bcur[1 - c] : 1 - V.bcur[c];
      launch_gs_wave_batch (K, V.L, V.B, V.color_blk[c], V.color_blk[c + 1], V.bxbuf (V.bcur[0]), V.bxbuf (V.bcur[1]), (int) V.rows0, V.bb, V.bxbuf (out), H.f32, st);
      V.bcur[c] = out;
      return;
   }
   launch_csr_spmv_batch (K, V.L, V.color_rb[c], V.color_rb[c + 1], V.bx, V.br, V.bb, 1, st);
   column_solves_batch (H, V, K, c, V.br, V.bx, 1, st);
}

static void ml_cycle_batch (MlHierarchy &H, int K, int l, hipStream_t st)
{
   MlLevel &V = H.lev[l];
   const int64_t nk = V.n * K;
   V.bcur[0] = V.bcur[1] = 0;
   if (l == (int) H.lev.size () - 1) {
      if (H.coarse_inv) {
         if (H.coarse_invf) launch_dense_matvec_f32_batch (K, H.coarse_invf, H.coarse_ldf, V.bb, V.bx, (int) V.n, st);
         else launch_dense_matvec_batch (K, H.coarse_inv, V.bb, V.bx, (int) V.n, st);
         return;
      }
      const int sweeps = H.tune->ml_coarsest_sweeps > 0 ? H.tune->ml_coarsest_sweeps : 30;
      launch_fill (V.bx, 0.0, nk, st);
      column_solves_batch (H, V, K, 0, V.bb, V.bx, 0, st);
      launch_csr_spmv_batch (K, V.L, V.color_rb[1], V.color_rb[2], V.bx, V.br, V.bb, 1, st);
      column_solves_batch (H, V, K, 1, V.br, V.bx, 1, st);
      for (int s = 1; s < sweeps; s++)
         for (int step = 0; step < 2; step++) {
            const int c = (s & 1) ? 1 - step : step;
            launch_csr_spmv_batch (K, V.L, V.color_rb[c], V.color_rb[c + 1], V.bx, V.br, V.bb, 1, st);
            column_solves_batch (H, V, K, c, V.br, V.bx, 1, st);
         }
      return;
   }
   launch_fill (V.bx, 0.0, nk, st);
   if (V.wave_fused) {
      column_solves_batch (H, V, K, 0, V.bb, V.bx2, 0, st);
      V.bcur[0] = 1;
      gs_half_batch (H, V, K, 1, st);
   } else {
      column_solves_batch (H, V, K, 0, V.bb, V.bx, 0, st);
      gs_half_batch (H, V, K, 1, st);
   }
   const int nu = (l >= H.coarse_from) ? H.nu_coarse : H.nu;
   for (int s = 1; s < nu; s++) { gs_half_batch (H, V, K, 0, st); gs_half_batch (H, V, K, 1, st); }
   MlLevel &C = H.lev[l + 1];
   const int gamma = (l >= H.gamma_from && l < H.gamma_to) ? 2 : 1;
   for (int g = 0; g < gamma; g++) {
      launch_csr_spmv_batch (K, V.L, 0, V.L.nrowblk, V.bxnow (), V.br, V.bb, 1, st);
      launch_restrict_sum_batch (K, V.rptr, V.ridx, V.br, C.bb, V.nc, st);
      ml_cycle_batch (H, K, l + 1, st);
      launch_prolong_add_batch (K, V.cmap, C.bxnow (), V.bxnow (), V.n, H.omega, st);
   }
   for (int s = 0; s < nu; s++) { gs_half_batch (H, V, K, 1, st); gs_half_batch (H, V, K, 0, st); }
}

void ml_apply_batch (MlHierarchy &H, int K, const double *r, double *z, hipStream_t st)
{
   MlLevel &V = H.lev[0];
   launch_gather_batch (K, H.perm0, r, V.bb, V.n, st);
   ml_cycle_batch (H, K, 0, st);
   launch_scatter_batch (K, H.perm0, V.bxnow (), z, V.n, st);
}

// the same from / to per-system vectors: src[k] = residual of system k (NULL: zeros), z = the K corrections interleaved,
// dst[k] (may be NULL) = a plain copy of column k
void ml_apply_batch_split (MlHierarchy &H, int K, const double *const *src, double *z, double *const *dst, hipStream_t st)
{
   MlLevel &V = H.lev[0];
   launch_gather_interleave (K, H.perm0, src, V.bb, V.n, st);
   ml_cycle_batch (H, K, 0, st);
   launch_scatter_split (K, H.perm0, V.bxnow (), z, dst, V.n, st);
}

// ================================================================ measurement helpers (bench.py, probes)
// one half sweep of level 0, colour 0: the residual rows (which = 0) or the column solves (which = 1)
void ml_time_piece (MlHierarchy &H, int which, hipStream_t st)
{
   MlLevel &V = H.lev[0];
   if (H.lev.size () < 2) return;
   if (which == 0) launch_csr_residual_range (V.L, V.color_rb[0], V.color_rb[1], V.x, V.b, V.r, st);
   else launch_colblock_apply_lanes (V.B, V.color_grp[0], V.color_grp[1], V.r, V.x, 1, st);
}

// compulsory HBM bytes (every array element counted once per kernel that must touch it):
//  which 0: residual rows of level 0, colour 0: its entries (value + column), row pointers, b in, r out, x once
//  which 1: column solves of level 0, colour 0: factors, r in, x in and out
//  which 2: one whole V(nu, nu) cycle
int64_t ml_bytes (const MlHierarchy &H, int which)
{
   if (H.lev.size () < 2) return 0;
   auto level_piece = [&] (const MlLevel &V, int colour, int what) -> int64_t {
      const int64_t rows = colour == 0 ? V.rows0 : V.n - V.rows0;
      const int64_t vb = V.L.valf ? 4 : 8, fb = V.B.fac_tf ? 4 : 8;
      if (what == 0) {
         // entries of the colour's rows: the colour-major CSR keeps them contiguous; split nnz by rows as an estimate is not
         // needed -- the host knows the exact count only at setup, so use the level's average row length
         const double per_row = V.n ? (double) V.L.nnz / (double) V.n : 0.0;
         return (int64_t) (per_row * (double) rows * (double) (vb + 4)) + rows * (4 + 8 + 8) + V.n * 8;
      }
      return rows * ((2 * V.B.P + 1) * fb + 8 + 8 + 8);
   };
   if (which == 0 || which == 1) return level_piece (H.lev[0], 0, which);
   int64_t total = 0;
   for (size_t l = 0; l + 1 < H.lev.size (); l++) {
      const MlLevel &V = H.lev[l];
      const int nu = ((int) l >= H.coarse_from) ? H.nu_coarse : H.nu;
      for (int c = 0; c < 2; c++) {
         total += (int64_t) (2 * nu) * level_piece (V, c, 1);                       // column solves: nu pre + nu post sweeps
         total += (int64_t) (2 * nu - (c == 0 ? 1 : 0)) * level_piece (V, c, 0);    // residual rows (the first half sweep needs none)
      }
      total += V.L.nnz * ((V.L.valf ? 4 : 8) + 4) + V.n * (4 + 8 + 8 + 8);         // full residual before the restriction
      total += V.n * (8 + 4) + V.nc * (8 + 4);                                      // restriction
      total += V.n * (8 + 8 + 4) + V.nc * 8;                                        // prolongation
   }
   const int64_t ncoarse = H.lev.back ().n;
   total += ncoarse * ncoarse * 8 + 2 * ncoarse * 8;                                // dense coarsest solve
   total += H.lev[0].n * (8 + 8 + 4) * 2;                                           // gather in, scatter out
   return total;
}
