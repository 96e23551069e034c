// Device-side construction of the multilevel hierarchy (round 3).
//
// What it replaces: the host threads that built the low-order twin, the connectivity-aware coarse cells, the Galerkin
// products and the colour-major operators of every level (multilevel.hip, rounds 1-2) -- the stand-in of SuperLU's
// symbolic + numeric factorisation (reference src/solve_ABglobal.c:349-360, src/SuperLU_brief_tree.txt:5-14), which at
// 1 degree cost five solves' worth of host time.  Here every row- and entry-level pass is a kernel; only column-level
// bookkeeping (10^4-10^5 items) goes through the host.
//
// Contract: the hierarchy is ENTRY FOR ENTRY the one the host routines of multilevel.hip build (which still serve the
// small levels and the no-GPU plan tests): same set numbering (a set's id is the rank of its lowest row), same tie rules,
// same summation order in every floating-point sum.  tests/test_gpu_setup.py compares the two.
//
// Shape of the kernels: one thread per row (or per coarse row, per set).  Rows of these operators hold 5-40 entries, so a
// thread's work is a short private segment of global memory; coalescing is poor, but the whole setup moves each array a
// handful of times and the alternative (LDS-tiled variants) is not worth its code until these show up in a profile.
#include "mlsetup.h"
#include "nkp_dev.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace mls {

#define MLS_T 256
#define CHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return (int) e_; } while (0)

static inline dim3 grid_for (int64_t n) { return dim3 ((unsigned) ((n + MLS_T - 1) / MLS_T)); }

// ---------------------------------------------------------------- exclusive scan (reduce, scan the block sums, scan)
#define SCAN_ITEMS 8
#define SCAN_BLOCK (MLS_T * SCAN_ITEMS)

__global__ __launch_bounds__ (MLS_T)
void scan_blocksum_kernel (const int *__restrict__ in, int64_t n, int *__restrict__ bsum)
{
   __shared__ int red[MLS_T];
   const int64_t base = (int64_t) blockIdx.x * SCAN_BLOCK;
   int s = 0;
#pragma unroll
   for (int u = 0; u < SCAN_ITEMS; u++) {
      const int64_t i = base + (int64_t) u * MLS_T + threadIdx.x;
      if (i < n) s += in[i];
   }
   red[threadIdx.x] = s;
   __syncthreads ();
   for (int off = MLS_T / 2; off > 0; off >>= 1) {
      if ((int) threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads ();
   }
   if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

// one workgroup: exclusive scan of bsum[0..nb) in place, bsum[nb] = total
__global__ __launch_bounds__ (MLS_T)
void scan_top_kernel (int *__restrict__ bsum, int nb)
{
   __shared__ int tmp[MLS_T];
   __shared__ int carry;
   if (threadIdx.x == 0) carry = 0;
   __syncthreads ();
   for (int b0 = 0; b0 < nb; b0 += MLS_T) {
      const int i = b0 + (int) threadIdx.x;
      const int v = i < nb ? bsum[i] : 0;
      tmp[threadIdx.x] = v;
      __syncthreads ();
      for (int off = 1; off < MLS_T; off <<= 1) {
         const int t = (int) threadIdx.x >= off ? tmp[threadIdx.x - off] : 0;
         __syncthreads ();
         tmp[threadIdx.x] += t;
         __syncthreads ();
      }
      const int incl = tmp[threadIdx.x];
      const int c = carry;
      if (i < nb) bsum[i] = c + incl - v;
      __syncthreads ();
      if (threadIdx.x == MLS_T - 1) carry = c + incl;
      __syncthreads ();
   }
   if (threadIdx.x == 0) bsum[nb] = carry;
}

__global__ __launch_bounds__ (MLS_T)
void scan_final_kernel (const int *__restrict__ in, int *__restrict__ out, int64_t n, const int *__restrict__ bsum, int nb)
{
   __shared__ int tsum[MLS_T];
   // thread t owns SCAN_ITEMS consecutive items, so the block-level scan is over the thread sums
   const int64_t base = (int64_t) blockIdx.x * SCAN_BLOCK + (int64_t) threadIdx.x * SCAN_ITEMS;
   int v[SCAN_ITEMS];
   int s = 0;
#pragma unroll
   for (int u = 0; u < SCAN_ITEMS; u++) {
      v[u] = (base + u < n) ? in[base + u] : 0;
      s += v[u];
   }
   tsum[threadIdx.x] = s;
   __syncthreads ();
   for (int off = 1; off < MLS_T; off <<= 1) {
      const int t = (int) threadIdx.x >= off ? tsum[threadIdx.x - off] : 0;
      __syncthreads ();
      tsum[threadIdx.x] += t;
      __syncthreads ();
   }
   int run = bsum[blockIdx.x] + tsum[threadIdx.x] - s;
#pragma unroll
   for (int u = 0; u < SCAN_ITEMS; u++) {
      if (base + u < n) out[base + u] = run;
      run += v[u];
   }
   if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}

int scan_exclusive (const int *d_in, int *d_out, int64_t n, hipStream_t st, int64_t *total_host)
{
   if (n <= 0) {
      CHK (hipMemsetAsync (d_out, 0, sizeof (int), st));
      if (total_host) { *total_host = 0; CHK (hipStreamSynchronize (st)); }
      return 0;
   }
   const int nb = (int) ((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
   DBuf<int> bsum;
   CHK (bsum.alloc ((size_t) nb + 1));
   hipLaunchKernelGGL (scan_blocksum_kernel, dim3 (nb), dim3 (MLS_T), 0, st, d_in, n, bsum.p);
   hipLaunchKernelGGL (scan_top_kernel, dim3 (1), dim3 (MLS_T), 0, st, bsum.p, nb);
   hipLaunchKernelGGL (scan_final_kernel, dim3 (nb), dim3 (MLS_T), 0, st, d_in, d_out, n, bsum.p, nb);
   int tot = 0;
   // the block sums are freed on return: wait for the kernels in any case
   if (total_host) CHK (hipMemcpyAsync (&tot, bsum.p + nb, sizeof (int), hipMemcpyDeviceToHost, st));
   CHK (hipStreamSynchronize (st));
   if (total_host) *total_host = tot;
   return (int) hipGetLastError ();
}

// ---------------------------------------------------------------- small helpers
__global__ void rows_to_cols_kernel (const int *__restrict__ blk_start, int ncol, int *__restrict__ col_of)
{
   const int c = blockIdx.x * MLS_T + threadIdx.x;
   if (c >= ncol) return;
   for (int r = blk_start[c]; r < blk_start[c + 1]; r++) col_of[r] = c;
}

int rows_to_cols (const int *d_blk_start, int ncol, int *d_col_of, hipStream_t st)
{
   if (ncol > 0) hipLaunchKernelGGL (rows_to_cols_kernel, grid_for (ncol), dim3 (MLS_T), 0, st, d_blk_start, ncol, d_col_of);
   return 0;
}

__global__ void fill_int_kernel (int *__restrict__ p, int v, int64_t n)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n) p[i] = v;
}

__global__ void iota_kernel (int *__restrict__ p, int64_t n)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n) p[i] = (int) i;
}

static void fill_int (int *p, int v, int64_t n, hipStream_t st)
{
   if (n > 0) hipLaunchKernelGGL (fill_int_kernel, grid_for (n), dim3 (MLS_T), 0, st, p, v, n);
}

__global__ void to_float_kernel (const double *__restrict__ src, float *__restrict__ dst, int64_t n)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n) dst[i] = (float) src[i];
}

int to_float (const double *d_src, float *d_dst, int64_t cnt, hipStream_t st)
{
   if (cnt > 0) hipLaunchKernelGGL (to_float_kernel, grid_for (cnt), dim3 (MLS_T), 0, st, d_src, d_dst, cnt);
   return 0;
}

// ---------------------------------------------------------------- low-order twin
// L = A + D - diag (rowsum D), D_ij = max (0, -a_ij, -a_ji) for i, j in different water columns (a_ji by bisection in the
// sorted row j); entries between columns whose coupling becomes exactly zero are not stored.  Row sums in stored order.
__global__ __launch_bounds__ (MLS_T)
void twin_count_kernel (int64_t n, const int *__restrict__ rowptr, const int *__restrict__ colind, const double *__restrict__ val,
                        const int *__restrict__ col_of, double *__restrict__ nv, int *__restrict__ keep)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i >= n) return;
   const int ci = col_of[i];
   double dsum = 0.0;
   int diag_pos = -1, cnt = 0;
   const int e1 = rowptr[i + 1];
   for (int e = rowptr[i]; e < e1; e++) {
      const int j = colind[e];
      double a = val[e];
      if (j == (int) i) { diag_pos = e; nv[e] = a; cnt++; continue; }
      const bool other = col_of[j] != ci;
      if (other) {
         int lo = rowptr[j], hi = rowptr[j + 1];
         const int end = hi;
         while (lo < hi) {                                  // first position with colind >= i
            const int mid = (lo + hi) >> 1;
            if (colind[mid] < (int) i) lo = mid + 1;
            else hi = mid;
         }
         const double aji = (lo < end && colind[lo] == (int) i) ? val[lo] : 0.0;
         double d = 0.0;
         if (-a > d) d = -a;
         if (-aji > d) d = -aji;
         a += d;
         dsum += d;
      }
      nv[e] = a;
      if (a != 0.0 || !other) cnt++;
   }
   if (diag_pos >= 0) nv[diag_pos] -= dsum;
   keep[i] = cnt;
}

__global__ __launch_bounds__ (MLS_T)
void twin_fill_kernel (int64_t n, const int *__restrict__ rowptr, const int *__restrict__ colind, const int *__restrict__ col_of,
                       const double *__restrict__ nv, const int *__restrict__ lrow, int *__restrict__ lcol, double *__restrict__ lval)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i >= n) return;
   const int ci = col_of[i];
   int q = lrow[i];
   const int e1 = rowptr[i + 1];
   for (int e = rowptr[i]; e < e1; e++) {
      const int j = colind[e];
      if (j != (int) i && col_of[j] != ci && nv[e] == 0.0) continue;
      lcol[q] = j;
      lval[q] = nv[e];
      q++;
   }
}

int twin (int64_t n, const int *d_rowptr, const int *d_colind, const double *d_val, const int *d_col_of, DevCsr &L, hipStream_t st)
{
   int nnz = 0;
   CHK (hipMemcpyAsync (&nnz, d_rowptr + n, sizeof (int), hipMemcpyDeviceToHost, st));
   CHK (hipStreamSynchronize (st));
   DBuf<double> nv;
   DBuf<int> lrow;
   CHK (nv.alloc ((size_t) nnz));
   CHK (lrow.alloc ((size_t) n + 1));
   hipLaunchKernelGGL (twin_count_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, d_rowptr, d_colind, d_val, d_col_of, nv.p, lrow.p);
   int64_t lnnz = 0;
   int rc = scan_exclusive (lrow.p, lrow.p, n, st, &lnnz);
   if (rc) return rc;
   DBuf<int> lcol;
   DBuf<double> lval;
   CHK (lcol.alloc ((size_t) lnnz));
   CHK (lval.alloc ((size_t) lnnz));
   hipLaunchKernelGGL (twin_fill_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, d_rowptr, d_colind, d_col_of, nv.p, lrow.p, lcol.p, lval.p);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   L.n = n;
   L.nnz = lnnz;
   L.rowptr = lrow.release ();
   L.colind = lcol.release ();
   L.val = lval.release ();
   return 0;
}

// ---------------------------------------------------------------- inverse of a many-to-one row map
__global__ void hist_kernel (const int *__restrict__ map, int64_t n, int *__restrict__ cnt)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n && map[i] >= 0) atomicAdd (&cnt[map[i]], 1);
}

__global__ void bucket_fill_kernel (const int *__restrict__ map, int64_t n, const int *__restrict__ ptr, int *__restrict__ cursor, int *__restrict__ items)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i >= n) return;
   const int k = map[i];
   if (k < 0) return;
   items[ptr[k] + atomicAdd (&cursor[k], 1)] = (int) i;
}

// every segment sorted ascending (segments are short: insertion sort in place)
__global__ void segment_sort_kernel (const int *__restrict__ ptr, int64_t nseg, int *__restrict__ items)
{
   const int64_t s = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (s >= nseg) return;
   const int b = ptr[s], e = ptr[s + 1];
   for (int k = b + 1; k < e; k++) {
      const int v = items[k];
      int q = k;
      while (q > b && items[q - 1] > v) { items[q] = items[q - 1]; q--; }
      items[q] = v;
   }
}

int inverse_map (const int *d_map, int64_t n, int64_t nc, int **d_rptr, int **d_ridx, hipStream_t st)
{
   DBuf<int> rptr, ridx, cursor;
   CHK (rptr.alloc ((size_t) nc + 1));
   CHK (ridx.alloc ((size_t) n));
   CHK (cursor.alloc ((size_t) nc));
   CHK (hipMemsetAsync (rptr.p, 0, ((size_t) nc + 1) * sizeof (int), st));
   CHK (hipMemsetAsync (cursor.p, 0, (size_t) (nc ? nc : 1) * sizeof (int), st));
   if (n > 0) hipLaunchKernelGGL (hist_kernel, grid_for (n), dim3 (MLS_T), 0, st, d_map, n, rptr.p);
   int rc = scan_exclusive (rptr.p, rptr.p, nc, st, nullptr);
   if (rc) return rc;
   if (n > 0) hipLaunchKernelGGL (bucket_fill_kernel, grid_for (n), dim3 (MLS_T), 0, st, d_map, n, rptr.p, cursor.p, ridx.p);
   if (nc > 0) hipLaunchKernelGGL (segment_sort_kernel, grid_for (nc), dim3 (MLS_T), 0, st, rptr.p, nc, ridx.p);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   *d_rptr = rptr.release ();
   *d_ridx = ridx.release ();
   return 0;
}

// ---------------------------------------------------------------- Galerkin product with a piecewise-constant P
__global__ void galerkin_bound_kernel (int64_t nc, const int *__restrict__ rptr, const int *__restrict__ ridx, const int *__restrict__ lrow, int *__restrict__ ub)
{
   const int64_t I = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (I >= nc) return;
   int s = 0;
   for (int q = rptr[I]; q < rptr[I + 1]; q++) { const int i = ridx[q]; s += lrow[i + 1] - lrow[i]; }
   ub[I] = s;
}

// One thread per coarse row: its fine rows in ascending order, their entries in stored order, accumulated into a list kept
// sorted by coarse column (the incoming columns are nearly ascending, so the search runs from the end).  acc[J] += v in
// exactly the order of the host routine; zeros are dropped except on the diagonal.
__global__ __launch_bounds__ (MLS_T)
void galerkin_accumulate_kernel (int64_t nc, const int *__restrict__ rptr, const int *__restrict__ ridx, const int *__restrict__ lrow,
                                 const int *__restrict__ lcol, const double *__restrict__ lval, const int *__restrict__ cmap,
                                 const int *__restrict__ ubptr, int *__restrict__ tJ, double *__restrict__ tV, int *__restrict__ cnt)
{
   const int64_t I = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (I >= nc) return;
   int *J = tJ + ubptr[I];
   double *V = tV + ubptr[I];
   int len = 0;
   for (int q = rptr[I]; q < rptr[I + 1]; q++) {
      const int i = ridx[q];
      const int e1 = lrow[i + 1];
      for (int e = lrow[i]; e < e1; e++) {
         const int c = cmap[lcol[e]];
         const double v = lval[e];
         int pos = len;
         while (pos > 0 && J[pos - 1] > c) pos--;
         if (pos > 0 && J[pos - 1] == c) V[pos - 1] += v;
         else {
            for (int k = len; k > pos; k--) { J[k] = J[k - 1]; V[k] = V[k - 1]; }
            J[pos] = c;
            V[pos] = 0.0 + v;
            len++;
         }
      }
   }
   int out = 0;
   for (int k = 0; k < len; k++)
      if (V[k] != 0.0 || J[k] == (int) I) { J[out] = J[k]; V[out] = V[k]; out++; }
   cnt[I] = out;
}

__global__ void galerkin_compact_kernel (int64_t nc, const int *__restrict__ ubptr, const int *__restrict__ tJ, const double *__restrict__ tV,
                                         const int *__restrict__ crow, int *__restrict__ ccol, double *__restrict__ cval)
{
   const int64_t I = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (I >= nc) return;
   const int b = crow[I], len = crow[I + 1] - b, src = ubptr[I];
   for (int k = 0; k < len; k++) { ccol[b + k] = tJ[src + k]; cval[b + k] = tV[src + k]; }
}

int galerkin (const DevCsr &L, const int *d_cmap, int64_t nc, DevCsr &C, hipStream_t st)
{
   int *rptr = nullptr, *ridx = nullptr;
   int rc = inverse_map (d_cmap, L.n, nc, &rptr, &ridx, st);
   if (rc) return rc;
   DBuf<int> rp, ri, ub, tJ, crow;
   DBuf<double> tV;
   rp.p = rptr; ri.p = ridx;
   CHK (ub.alloc ((size_t) nc + 1));
   CHK (crow.alloc ((size_t) nc + 1));
   if (nc > 0) hipLaunchKernelGGL (galerkin_bound_kernel, grid_for (nc), dim3 (MLS_T), 0, st, nc, rptr, ridx, L.rowptr, ub.p);
   int64_t total = 0;
   if ((rc = scan_exclusive (ub.p, ub.p, nc, st, &total))) return rc;
   CHK (tJ.alloc ((size_t) total));
   CHK (tV.alloc ((size_t) total));
   if (nc > 0) hipLaunchKernelGGL (galerkin_accumulate_kernel, grid_for (nc), dim3 (MLS_T), 0, st, nc, rptr, ridx, L.rowptr, L.colind, L.val, d_cmap, ub.p, tJ.p, tV.p, crow.p);
   int64_t cnnz = 0;
   if ((rc = scan_exclusive (crow.p, crow.p, nc, st, &cnnz))) return rc;
   DBuf<int> ccol;
   DBuf<double> cval;
   CHK (ccol.alloc ((size_t) cnnz));
   CHK (cval.alloc ((size_t) cnnz));
   if (nc > 0) hipLaunchKernelGGL (galerkin_compact_kernel, grid_for (nc), dim3 (MLS_T), 0, st, nc, ub.p, tJ.p, tV.p, crow.p, ccol.p, cval.p);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   C.n = nc;
   C.nnz = cnnz;
   C.rowptr = crow.release ();
   C.colind = ccol.release ();
   C.val = cval.release ();
   return 0;
}

// ---------------------------------------------------------------- colour-major operator
__global__ void perm_len_kernel (int64_t n, const int *__restrict__ perm, const int *__restrict__ lrow, int *__restrict__ len)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n) { const int o = perm[i]; len[i] = lrow[o + 1] - lrow[o]; }
}

__global__ __launch_bounds__ (MLS_T)
void permute_rows_kernel (int64_t n, const int *__restrict__ perm, const int *__restrict__ inv, const int *__restrict__ lrow,
                          const int *__restrict__ lcol, const double *__restrict__ lval, const int *__restrict__ prow,
                          int *__restrict__ pcol, double *__restrict__ pval)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i >= n) return;
   const int o = perm[i];
   const int b = prow[i];
   int len = 0;
   for (int e = lrow[o]; e < lrow[o + 1]; e++) {
      const int c = inv[lcol[e]];
      const double v = lval[e];
      int q = b + len;
      while (q > b && pcol[q - 1] > c) { pcol[q] = pcol[q - 1]; pval[q] = pval[q - 1]; q--; }
      pcol[q] = c;
      pval[q] = v;
      len++;
   }
}

int permute_operator (const DevCsr &L, const int *d_perm, const int *d_inv, int **d_prow, int **d_pcol, double **d_pval, int64_t pad, hipStream_t st)
{
   DBuf<int> prow, pcol;
   DBuf<double> pval;
   CHK (prow.alloc ((size_t) L.n + 1));
   CHK (pcol.alloc ((size_t) (L.nnz + pad)));
   CHK (pval.alloc ((size_t) (L.nnz + pad)));
   if (pad > 0) {
      CHK (hipMemsetAsync (pcol.p + L.nnz, 0, (size_t) pad * sizeof (int), st));
      CHK (hipMemsetAsync (pval.p + L.nnz, 0, (size_t) pad * sizeof (double), st));
   }
   if (L.n > 0) hipLaunchKernelGGL (perm_len_kernel, grid_for (L.n), dim3 (MLS_T), 0, st, L.n, d_perm, L.rowptr, prow.p);
   int rc = scan_exclusive (prow.p, prow.p, L.n, st, nullptr);
   if (rc) return rc;
   if (L.n > 0) hipLaunchKernelGGL (permute_rows_kernel, grid_for (L.n), dim3 (MLS_T), 0, st, L.n, d_perm, d_inv, L.rowptr, L.colind, L.val, prow.p, pcol.p, pval.p);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   *d_prow = prow.release ();
   *d_pcol = pcol.release ();
   *d_pval = pval.release ();
   return 0;
}

__global__ void colour_major_kernel (int64_t n, const int *__restrict__ blk_start, const int *__restrict__ col_of, const int *__restrict__ newstart,
                                     int *__restrict__ perm, int *__restrict__ inv)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= n) return;
   const int c = col_of[r];
   const int i = newstart[c] + ((int) r - blk_start[c]);
   inv[r] = i;
   perm[i] = (int) r;
}

int colour_major_maps (const int *d_blk_start, const int *d_col_of, const int *d_newstart, int64_t n, int *d_perm, int *d_inv, hipStream_t st)
{
   if (n > 0) hipLaunchKernelGGL (colour_major_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, d_blk_start, d_col_of, d_newstart, d_perm, d_inv);
   return 0;
}

__global__ void permuted_cmap_kernel (int64_t n, const int *__restrict__ cmap, const int *__restrict__ perm, const int *__restrict__ inv_c, int *__restrict__ out)
{
   const int64_t i = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (i < n) out[i] = inv_c[cmap[perm[i]]];
}

int permuted_cmap (const int *d_cmap, const int *d_perm, const int *d_inv_coarse, int64_t n, int *d_out, hipStream_t st)
{
   if (n > 0) hipLaunchKernelGGL (permuted_cmap_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, d_cmap, d_perm, d_inv_coarse, d_out);
   return 0;
}

// ---------------------------------------------------------------- CSR-stream row blocks
// The greedy rule of build_rowblocks_host (spmv.hip) run independently on chunks of RB_CHUNK rows, one thread per chunk:
// where a block ends changes nothing in the result of an SpMV (every row is summed in stored order by one lane), so the
// forced boundary per chunk only costs one short block in ~25.
#define RB_CHUNK 4096

__global__ void rowblock_chunk_kernel (const int *__restrict__ rowptr, int64_t r0, int64_t r1, int nchunk, int *__restrict__ tmp, int *__restrict__ cnt)
{
   const int c = blockIdx.x * MLS_T + threadIdx.x;
   if (c >= nchunk) return;
   int64_t r = r0 + (int64_t) c * RB_CHUNK;
   const int64_t rend = min (r + (int64_t) RB_CHUNK, r1);
   int *out = tmp + (int64_t) c * RB_CHUNK;
   int k = 0;
   while (r < rend) {
      int64_t e = r + 1;
      const int64_t base = rowptr[r];
      while (e < rend && (e - r) < NKP_SPMV_MAX_ROWS && (int64_t) rowptr[e + 1] - base <= NKP_SPMV_LDS_NNZ) e++;
      out[k++] = (int) r;             // first row of the block
      r = e;
   }
   cnt[c] = k;
}

__global__ void rowblock_gather_kernel (int nchunk, const int *__restrict__ tmp, const int *__restrict__ ptr, int *__restrict__ out, int last)
{
   const int c = blockIdx.x * MLS_T + threadIdx.x;
   if (c >= nchunk) return;
   const int b = ptr[c], k = ptr[c + 1] - b;
   const int *src = tmp + (int64_t) c * RB_CHUNK;
   for (int q = 0; q < k; q++) out[b + q] = src[q];
   if (c == nchunk - 1) out[b + k] = last;
}

int row_blocks (const int *d_rowptr, int64_t r0, int64_t r1, int **d_out, int *nblocks, hipStream_t st)
{
   *d_out = nullptr;
   *nblocks = 0;
   if (r1 <= r0) return 0;
   const int nchunk = (int) ((r1 - r0 + RB_CHUNK - 1) / RB_CHUNK);
   DBuf<int> tmp, cnt, out;
   CHK (tmp.alloc ((size_t) nchunk * RB_CHUNK));
   CHK (cnt.alloc ((size_t) nchunk + 1));
   hipLaunchKernelGGL (rowblock_chunk_kernel, grid_for (nchunk), dim3 (MLS_T), 0, st, d_rowptr, r0, r1, nchunk, tmp.p, cnt.p);
   int64_t total = 0;
   int rc = scan_exclusive (cnt.p, cnt.p, nchunk, st, &total);
   if (rc) return rc;
   CHK (out.alloc ((size_t) total + 1));
   hipLaunchKernelGGL (rowblock_gather_kernel, grid_for (nchunk), dim3 (MLS_T), 0, st, nchunk, tmp.p, cnt.p, out.p, (int) r1);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   *d_out = out.release ();
   *nblocks = (int) total;
   return 0;
}

// ================================================================ connectivity-aware coarse cells
// multilevel.hip: split_aggregate, pass for pass.  depth (r) = ktop[col] + (r - blk_start[col]).
struct AggDev {
   int64_t n;
   int ncol;
   const int *rowptr, *colind;
   const double *val;
   const int *blk_start, *col_of, *ktop, *group, *col_t;
   const unsigned char *dang;
};

__device__ __forceinline__ int agg_depth (const AggDev &a, int r) { const int c = a.col_of[r]; return a.ktop[c] + (r - a.blk_start[c]); }
__device__ __forceinline__ int agg_row_at (const AggDev &a, int c, int k)
{
   const int r = a.blk_start[c] + (k - a.ktop[c]);
   return (k >= a.ktop[c] && r < a.blk_start[c + 1]) ? r : -1;
}

// lock-free union-find: a root is only ever linked under a LOWER index, so the final root of a set is its lowest row
__device__ __forceinline__ int uf_load (const int *p, int x) { return __hip_atomic_load (&p[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ int uf_find (int *p, int x)
{
   for (;;) {
      const int px = uf_load (p, x);
      if (px == x) return x;
      const int gp = uf_load (p, px);
      if (gp != px) __hip_atomic_store (&p[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // path halving: any ancestor is a valid parent
      x = px;
   }
}

__device__ void uf_unite (int *p, int a, int b)
{
   for (;;) {
      a = uf_find (p, a);
      b = uf_find (p, b);
      if (a == b) return;
      if (a > b) { const int t = a; a = b; b = t; }
      if (atomicCAS (&p[b], b, a) == b) return;
   }
}

__global__ void agg_diag_kernel (AggDev a, double *__restrict__ diag)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   double d = 0.0;
   for (int e = a.rowptr[r]; e < a.rowptr[r + 1]; e++)
      if (a.colind[e] == (int) r) d = fabs (a.val[e]);
   diag[r] = d;
}

// felt[c2] = max over rows outside c2 (same tracer, one end a stub) of |a| / |diag of that row|;  best[c] = the column's
// strongest such coupling.  Non-negative doubles order like their bit patterns, so both maxima are integer atomics.
__global__ void agg_stub_max_kernel (AggDev a, const double *__restrict__ diag, unsigned long long *__restrict__ felt, unsigned long long *__restrict__ best)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   const int c = a.col_of[r];
   const double dg = diag[r];
   for (int e = a.rowptr[r]; e < a.rowptr[r + 1]; e++) {
      const int j = a.colind[e], c2 = a.col_of[j];
      if (c2 == c || a.col_t[c2] != a.col_t[c]) continue;
      if (a.ktop[c2] == 0 && a.ktop[c] == 0) continue;
      const double v = fabs (a.val[e]);
      const double f = dg > 0.0 ? v / dg : 1.0e300;
      atomicMax (&felt[c2], (unsigned long long) __double_as_longlong (f));
      atomicMax (&best[c], (unsigned long long) __double_as_longlong (v));
   }
}

// the LAST entry (rows ascending, entries ascending) that attains the column's maximum is its anchor
__global__ void agg_stub_anchor_kernel (AggDev a, const unsigned long long *__restrict__ best, int *__restrict__ pos)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   const int c = a.col_of[r];
   for (int e = a.rowptr[r]; e < a.rowptr[r + 1]; e++) {
      const int j = a.colind[e], c2 = a.col_of[j];
      if (c2 == c || a.col_t[c2] != a.col_t[c]) continue;
      if (a.ktop[c2] == 0 && a.ktop[c] == 0) continue;
      const double v = fabs (a.val[e]);
      if ((unsigned long long) __double_as_longlong (v) == best[c]) atomicMax (&pos[c], e);
   }
}

__global__ void agg_anchor_row_kernel (int ncol, const int *__restrict__ pos, const int *__restrict__ colind, int *__restrict__ anchor)
{
   const int c = blockIdx.x * MLS_T + threadIdx.x;
   if (c < ncol) anchor[c] = pos[c] >= 0 ? colind[pos[c]] : -1;
}

// PASS 0: every lateral edge -> U0 (if pockets are merged), edges inside a group -> U
// PASS 1: rows of small same-depth sets (pockets): their edges ACROSS groups -> U
template <int PASS>
__global__ __launch_bounds__ (MLS_T)
void agg_edges_kernel (AggDev a, int *__restrict__ U, int *__restrict__ U0, const int *__restrict__ root0, const int *__restrict__ size0, int pocket)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   if (PASS == 1) { const int s = size0[root0[r]]; if (!(s <= pocket && s > 1)) return; }
   const int c = a.col_of[r];
   if (a.dang[c]) return;
   const int k = a.ktop[c] + ((int) r - a.blk_start[c]);
   const int tc = a.col_t[c], gc = a.group[c];
   for (int e = a.rowptr[r]; e < a.rowptr[r + 1]; e++) {
      const int j = a.colind[e], c2 = a.col_of[j];
      if (c2 == c || a.dang[c2] || a.col_t[c2] != tc) continue;
      const int dk = (a.ktop[c2] + (j - a.blk_start[c2])) - k;
      if (dk < -1 || dk > 1) continue;
      const int t = agg_row_at (a, c2, k);
      if (t < 0) continue;
      const bool same = a.group[c2] == gc;
      if (PASS == 0) {
         if (pocket > 0) uf_unite (U0, (int) r, t);
         if (same) uf_unite (U, (int) r, t);
      } else if (!same)
         uf_unite (U, (int) r, t);
   }
}

__global__ void agg_roots_kernel (int64_t n, int *__restrict__ U, int *__restrict__ root, int *__restrict__ size /* may be NULL */, int *__restrict__ isroot /* may be NULL */)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= n) return;
   const int q = uf_find (U, (int) r);
   root[r] = q;
   if (size) atomicAdd (&size[q], 1);
   if (isroot) isroot[r] = q == (int) r ? 1 : 0;
}

// sets numbered by their lowest row (= their root); comp[r] = id of r's set; kcomp[id] = depth of the set
__global__ void agg_comp_kernel (AggDev a, const int *__restrict__ root, const int *__restrict__ rank_of_row, int *__restrict__ comp, int *__restrict__ kcomp)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   const int id = rank_of_row[root[r]];
   comp[r] = id;
   if (root[r] == (int) r) kcomp[id] = agg_depth (a, (int) r);
}

// parent set of the pair (row r, row below r in the same column); -1 for the last row of a column
__global__ void agg_pair_parent_kernel (AggDev a, const int *__restrict__ comp, int *__restrict__ par)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   const int c = a.col_of[r];
   par[r] = ((int) r + 1 < a.blk_start[c + 1]) ? comp[r] : -1;
}

// one thread per parent set: its rows' children sorted, runs counted.  bestchi[par] = child with the largest overlap (ties:
// lowest id); bestpar[chi] = parent with the largest overlap (ties: lowest id) through a packed 64-bit maximum.
__global__ void agg_overlap_kernel (int ncomp, const int *__restrict__ bptr, int *__restrict__ items, const int *__restrict__ comp,
                                    int *__restrict__ bestchi, unsigned long long *__restrict__ bestpar_packed)
{
   const int par = blockIdx.x * MLS_T + threadIdx.x;
   if (par >= ncomp) return;
   const int b = bptr[par], e = bptr[par + 1];
   for (int k = b; k < e; k++) items[k] = comp[items[k] + 1];       // row -> set of the row below it
   for (int k = b + 1; k < e; k++) {
      const int v = items[k];
      int q = k;
      while (q > b && items[q - 1] > v) { items[q] = items[q - 1]; q--; }
      items[q] = v;
   }
   int bc = -1, bcnt = 0;
   for (int k = b; k < e;) {
      int k2 = k;
      while (k2 < e && items[k2] == items[k]) k2++;
      const int chi = items[k], cnt = k2 - k;
      if (cnt > bcnt) { bcnt = cnt; bc = chi; }
      atomicMax (&bestpar_packed[chi], ((unsigned long long) (unsigned) cnt << 32) | (unsigned long long) (0xFFFFFFFFu - (unsigned) par));
      k = k2;
   }
   bestchi[par] = bc;
}

// head of a coarse column: a set that does not continue its best parent's column
__global__ void agg_heads_kernel (int ncomp, const unsigned long long *__restrict__ bestpar_packed, const int *__restrict__ bestchi,
                                  int *__restrict__ link, int *__restrict__ ishead)
{
   const int id = blockIdx.x * MLS_T + threadIdx.x;
   if (id >= ncomp) return;
   const unsigned long long pk = bestpar_packed[id];
   const int par = pk ? (int) (0xFFFFFFFFu - (unsigned) (pk & 0xFFFFFFFFull)) : -1;
   const bool cont = par >= 0 && bestchi[par] == id;
   link[id] = cont ? par : id;
   ishead[id] = cont ? 0 : 1;
}

__global__ void agg_head_list_kernel (int ncomp, const int *__restrict__ ishead, const int *__restrict__ hpos, const int *__restrict__ kcomp,
                                      int *__restrict__ head_id, int *__restrict__ head_k)
{
   const int id = blockIdx.x * MLS_T + threadIdx.x;
   if (id >= ncomp || !ishead[id]) return;
   head_id[hpos[id]] = id;
   head_k[hpos[id]] = kcomp[id];
}

__global__ void agg_jump_kernel (int ncomp, int *__restrict__ link)
{
   const int id = blockIdx.x * MLS_T + threadIdx.x;
   if (id >= ncomp) return;
   const int l = link[id];
   const int l2 = link[l];
   if (l2 != l) link[id] = l2;          // in place: any ancestor on the way to the head is a valid link
}

// raw coarse column of every set; sets per column
__global__ void agg_ccol_kernel (int ncomp, const int *__restrict__ link, const int *__restrict__ hpos, const int *__restrict__ col_of_head,
                                 int *__restrict__ ccol, int *__restrict__ cc_len)
{
   const int id = blockIdx.x * MLS_T + threadIdx.x;
   if (id >= ncomp) return;
   const int q = col_of_head[hpos[link[id]]];
   ccol[id] = q;
   atomicAdd (&cc_len[q], 1);
}

__global__ void agg_minrow_kernel (AggDev a, const int *__restrict__ comp, const int *__restrict__ ccol, int *__restrict__ minrow)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   if (a.dang[a.col_of[r]]) return;
   atomicMin (&minrow[ccol[comp[r]]], (int) r);
}

__global__ void agg_cmap_kernel (AggDev a, const int *__restrict__ comp, const int *__restrict__ ccol, const int *__restrict__ newid,
                                 const int *__restrict__ cblk, const int *__restrict__ cktop, int *__restrict__ cmap)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   if (a.dang[a.col_of[r]]) { cmap[r] = -1; return; }
   const int q = newid[ccol[comp[r]]];
   cmap[r] = cblk[q] + (agg_depth (a, (int) r) - cktop[q]);
}

__global__ void agg_absorb_kernel (AggDev a, const int *__restrict__ anchor, int *__restrict__ cmap)
{
   const int64_t r = (int64_t) blockIdx.x * MLS_T + threadIdx.x;
   if (r >= a.n) return;
   const int c = a.col_of[r];
   if (a.dang[c]) cmap[r] = cmap[anchor[c]];       // the anchor's column is never absorbed itself
}

template <class T>
static int upload_vec (DBuf<T> &d, const T *src, size_t cnt, hipStream_t st)
{
   CHK (d.alloc (cnt));
   if (cnt) CHK (hipMemcpyAsync (d.p, src, cnt * sizeof (T), hipMemcpyHostToDevice, st));
   return 0;
}

int aggregate (const AggregateIn &in, AggregateOut &out, hipStream_t st)
{
   const int64_t n = in.n;
   const int ncol = in.ncol;
   int rc;
   DBuf<int> d_group, d_col_t;
   DBuf<unsigned char> d_dang;
   if ((rc = upload_vec (d_group, in.h_group, (size_t) ncol, st)) || (rc = upload_vec (d_col_t, in.h_col_t, (size_t) ncol, st))) return rc;
   CHK (d_dang.alloc ((size_t) ncol));
   CHK (hipMemsetAsync (d_dang.p, 0, (size_t) (ncol ? ncol : 1), st));
   AggDev a = { n, ncol, in.rowptr, in.colind, in.val, in.blk_start, in.col_of, in.ktop, d_group.p, d_col_t.p, d_dang.p };

   // ---- leaf stubs (columns that start below the surface and that no outside row feels)
   bool have_stubs = false;
   for (int c = 0; c < ncol && !have_stubs; c++) have_stubs = in.h_ktop[c] > 0;
   std::vector<unsigned char> dang ((size_t) ncol, 0);
   std::vector<int> anchor ((size_t) ncol, -1);
   DBuf<int> d_anchor;
   CHK (d_anchor.alloc ((size_t) ncol));
   if (have_stubs && in.tau > 0.0) {
      DBuf<double> diag;
      DBuf<unsigned long long> felt, best;
      DBuf<int> pos;
      CHK (diag.alloc ((size_t) n));
      CHK (felt.alloc ((size_t) ncol));
      CHK (best.alloc ((size_t) ncol));
      CHK (pos.alloc ((size_t) ncol));
      CHK (hipMemsetAsync (felt.p, 0, (size_t) ncol * sizeof (unsigned long long), st));
      CHK (hipMemsetAsync (best.p, 0, (size_t) ncol * sizeof (unsigned long long), st));
      fill_int (pos.p, -1, ncol, st);
      hipLaunchKernelGGL (agg_diag_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, diag.p);
      hipLaunchKernelGGL (agg_stub_max_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, diag.p, felt.p, best.p);
      hipLaunchKernelGGL (agg_stub_anchor_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, best.p, pos.p);
      hipLaunchKernelGGL (agg_anchor_row_kernel, grid_for (ncol), dim3 (MLS_T), 0, st, ncol, pos.p, in.colind, d_anchor.p);
      std::vector<unsigned long long> h_felt ((size_t) ncol);
      CHK (hipMemcpyAsync (h_felt.data (), felt.p, (size_t) ncol * sizeof (unsigned long long), hipMemcpyDeviceToHost, st));
      CHK (hipMemcpyAsync (anchor.data (), d_anchor.p, (size_t) ncol * sizeof (int), hipMemcpyDeviceToHost, st));
      CHK (hipStreamSynchronize (st));
      // column of a row, on the host: bisection in blk_start
      auto col_of_row = [&] (int r) { return (int) (std::upper_bound (in.h_blk_start, in.h_blk_start + ncol + 1, r) - in.h_blk_start) - 1; };
      for (int c = 0; c < ncol; c++) {
         double f;
         memcpy (&f, &h_felt[(size_t) c], sizeof f);
         dang[(size_t) c] = (in.h_ktop[c] > 0 && f < in.tau && anchor[(size_t) c] >= 0) ? 1 : 0;
      }
      std::vector<unsigned char> bad ((size_t) ncol, 0);
      for (int c = 0; c < ncol; c++) bad[(size_t) c] = dang[(size_t) c] && dang[(size_t) col_of_row (anchor[(size_t) c])];
      for (int c = 0; c < ncol; c++)
         if (bad[(size_t) c]) dang[(size_t) c] = 0;
      CHK (hipMemcpyAsync (d_dang.p, dang.data (), (size_t) ncol, hipMemcpyHostToDevice, st));
   }

   // ---- lateral edges between cells of the same depth
   DBuf<int> U, U0, root, root0, size0;
   CHK (U.alloc ((size_t) n));
   CHK (root.alloc ((size_t) n));
   hipLaunchKernelGGL (iota_kernel, grid_for (n), dim3 (MLS_T), 0, st, U.p, n);
   if (in.pocket > 0) {
      CHK (U0.alloc ((size_t) n));
      CHK (root0.alloc ((size_t) n));
      CHK (size0.alloc ((size_t) n));
      hipLaunchKernelGGL (iota_kernel, grid_for (n), dim3 (MLS_T), 0, st, U0.p, n);
      CHK (hipMemsetAsync (size0.p, 0, (size_t) n * sizeof (int), st));
   }
   hipLaunchKernelGGL (agg_edges_kernel<0>, grid_for (n), dim3 (MLS_T), 0, st, a, U.p, U0.p, (const int *) nullptr, (const int *) nullptr, in.pocket);
   if (in.pocket > 0) {
      hipLaunchKernelGGL (agg_roots_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, U0.p, root0.p, size0.p, (int *) nullptr);
      hipLaunchKernelGGL (agg_edges_kernel<1>, grid_for (n), dim3 (MLS_T), 0, st, a, U.p, U0.p, (const int *) root0.p, (const int *) size0.p, in.pocket);
   }
   // ---- sets numbered by their lowest row
   DBuf<int> rank_of_row, comp, kcomp;
   CHK (rank_of_row.alloc ((size_t) n + 1));
   CHK (comp.alloc ((size_t) n));
   hipLaunchKernelGGL (agg_roots_kernel, grid_for (n), dim3 (MLS_T), 0, st, n, U.p, root.p, (int *) nullptr, rank_of_row.p);
   int64_t ncomp64 = 0;
   if ((rc = scan_exclusive (rank_of_row.p, rank_of_row.p, n, st, &ncomp64))) return rc;
   const int ncomp = (int) ncomp64;
   U0.reset (); root0.reset (); size0.reset ();
   CHK (kcomp.alloc ((size_t) ncomp));
   hipLaunchKernelGGL (agg_comp_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, root.p, rank_of_row.p, comp.p, kcomp.p);
   // ---- overlaps between a set and the sets directly below it
   DBuf<int> bestchi, link, ishead, hpos;
   DBuf<unsigned long long> bestpar;
   CHK (bestchi.alloc ((size_t) ncomp));
   CHK (bestpar.alloc ((size_t) ncomp));
   CHK (hipMemsetAsync (bestpar.p, 0, (size_t) (ncomp ? ncomp : 1) * sizeof (unsigned long long), st));
   {
      DBuf<int> par;
      CHK (par.alloc ((size_t) n));
      hipLaunchKernelGGL (agg_pair_parent_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, comp.p, par.p);
      int *bptr = nullptr, *items = nullptr;
      if ((rc = inverse_map (par.p, n, ncomp, &bptr, &items, st))) return rc;
      DBuf<int> bp, it;
      bp.p = bptr; it.p = items;
      if (ncomp > 0) hipLaunchKernelGGL (agg_overlap_kernel, grid_for (ncomp), dim3 (MLS_T), 0, st, ncomp, bptr, items, comp.p, bestchi.p, bestpar.p);
      CHK (hipStreamSynchronize (st));
   }
   // ---- threading through depth: heads open coarse columns, numbered in order of (depth, id)
   CHK (link.alloc ((size_t) ncomp));
   CHK (ishead.alloc ((size_t) ncomp));
   CHK (hpos.alloc ((size_t) ncomp + 1));
   if (ncomp > 0) hipLaunchKernelGGL (agg_heads_kernel, grid_for (ncomp), dim3 (MLS_T), 0, st, ncomp, bestpar.p, bestchi.p, link.p, ishead.p);
   int64_t nraw64 = 0;
   if ((rc = scan_exclusive (ishead.p, hpos.p, ncomp, st, &nraw64))) return rc;
   const int nraw = (int) nraw64;
   DBuf<int> head_id, head_k;
   CHK (head_id.alloc ((size_t) nraw));
   CHK (head_k.alloc ((size_t) nraw));
   if (ncomp > 0) hipLaunchKernelGGL (agg_head_list_kernel, grid_for (ncomp), dim3 (MLS_T), 0, st, ncomp, ishead.p, hpos.p, kcomp.p, head_id.p, head_k.p);
   std::vector<int> h_head_k ((size_t) nraw), col_of_head ((size_t) nraw), cc_ktop ((size_t) nraw);
   CHK (hipMemcpyAsync (h_head_k.data (), head_k.p, (size_t) nraw * sizeof (int), hipMemcpyDeviceToHost, st));
   CHK (hipStreamSynchronize (st));
   {
      // stable counting sort of the heads (ascending id) by depth
      int kmax = 0;
      for (int q = 0; q < nraw; q++) kmax = std::max (kmax, h_head_k[(size_t) q]);
      std::vector<int> kptr ((size_t) kmax + 2, 0);
      for (int q = 0; q < nraw; q++) kptr[(size_t) h_head_k[(size_t) q] + 1]++;
      for (int k = 0; k <= kmax; k++) kptr[(size_t) k + 1] += kptr[(size_t) k];
      for (int q = 0; q < nraw; q++) {
         const int col = kptr[(size_t) h_head_k[(size_t) q]]++;
         col_of_head[(size_t) q] = col;
         cc_ktop[(size_t) col] = h_head_k[(size_t) q];
      }
      // a chain runs from its head to the deepest cell of its column at most: pointer jumping halves it per round
      int deepest = 0;
      for (int c = 0; c < ncol; c++) deepest = std::max (deepest, in.h_ktop[c] + (in.h_blk_start[c + 1] - in.h_blk_start[c]));
      int rounds = 1;
      while ((1 << rounds) < deepest + 2) rounds++;
      for (int q = 0; q < rounds && ncomp > 0; q++) hipLaunchKernelGGL (agg_jump_kernel, grid_for (ncomp), dim3 (MLS_T), 0, st, ncomp, link.p);
   }
   DBuf<int> d_col_of_head, ccol, cc_len, minrow;
   if ((rc = upload_vec (d_col_of_head, col_of_head.data (), (size_t) nraw, st))) return rc;
   CHK (ccol.alloc ((size_t) ncomp));
   CHK (cc_len.alloc ((size_t) nraw));
   CHK (minrow.alloc ((size_t) nraw));
   CHK (hipMemsetAsync (cc_len.p, 0, (size_t) (nraw ? nraw : 1) * sizeof (int), st));
   fill_int (minrow.p, 2147483647, nraw, st);
   if (ncomp > 0) hipLaunchKernelGGL (agg_ccol_kernel, grid_for (ncomp), dim3 (MLS_T), 0, st, ncomp, link.p, hpos.p, d_col_of_head.p, ccol.p, cc_len.p);
   hipLaunchKernelGGL (agg_minrow_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, comp.p, ccol.p, minrow.p);
   std::vector<int> h_len ((size_t) nraw), h_minrow ((size_t) nraw);
   CHK (hipMemcpyAsync (h_len.data (), cc_len.p, (size_t) nraw * sizeof (int), hipMemcpyDeviceToHost, st));
   CHK (hipMemcpyAsync (h_minrow.data (), minrow.p, (size_t) nraw * sizeof (int), hipMemcpyDeviceToHost, st));
   CHK (hipStreamSynchronize (st));
   // ---- coarse columns (host, column-level): absorbed stubs own no coarse column
   std::vector<int> newid ((size_t) nraw, -1);
   int ncc = 0;
   for (int q = 0; q < nraw; q++)
      if (h_minrow[(size_t) q] != 2147483647) newid[(size_t) q] = ncc++;
   out.blk_start.assign ((size_t) ncc + 1, 0);
   out.ktop.resize ((size_t) ncc);
   out.group.resize ((size_t) ncc);
   out.stubs = out.absorbed = 0;
   for (int q = 0; q < nraw; q++) {
      const int id = newid[(size_t) q];
      if (id < 0) continue;
      out.blk_start[(size_t) id + 1] = h_len[(size_t) q];
      out.ktop[(size_t) id] = cc_ktop[(size_t) q];
      const int r = h_minrow[(size_t) q];
      const int c = (int) (std::upper_bound (in.h_blk_start, in.h_blk_start + ncol + 1, r) - in.h_blk_start) - 1;
      out.group[(size_t) id] = in.h_group[c];
      if (cc_ktop[(size_t) q] > 0) out.stubs++;
   }
   for (int q = 0; q < ncc; q++) out.blk_start[(size_t) q + 1] += out.blk_start[(size_t) q];
   for (int c = 0; c < ncol; c++) out.absorbed += dang[(size_t) c] ? 1 : 0;
   DBuf<int> d_newid, d_cblk, d_cktop, cmap;
   if ((rc = upload_vec (d_newid, newid.data (), (size_t) nraw, st)) || (rc = upload_vec (d_cblk, out.blk_start.data (), (size_t) ncc + 1, st)) ||
       (rc = upload_vec (d_cktop, out.ktop.data (), (size_t) ncc, st)))
      return rc;
   CHK (cmap.alloc ((size_t) n));
   hipLaunchKernelGGL (agg_cmap_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, comp.p, ccol.p, d_newid.p, d_cblk.p, d_cktop.p, cmap.p);
   if (out.absorbed) hipLaunchKernelGGL (agg_absorb_kernel, grid_for (n), dim3 (MLS_T), 0, st, a, d_anchor.p, cmap.p);
   CHK (hipStreamSynchronize (st));
   CHK (hipGetLastError ());
   out.cmap = cmap.release ();
   return 0;
}

}  // namespace mls
