// K right-hand sides through ONE sweep of the matrix: the operator and the multilevel cycle on K interleaved vectors.
//
// The reference solves its tracers one after the other against one factorisation (RHS loop, reference
// src/solve_ABglobal.c:370-409).  Every kernel of a solve here is bound by the matrix / factor streams (12 bytes per entry of
// A, 8 per entry of a level operator, 20 per row of column factors) or, on the small levels, by launch latency -- both are
// per SWEEP, not per right-hand side.  So K systems share the sweeps: vectors are interleaved, X[i * K + k] = row i of system
// k (K = 2 or 4: one or two 16-byte loads per gathered row), and every kernel below does for the K columns exactly what its
// single-vector twin (spmv.hip, colblock.hip, blas1.hip) does for one -- same products, same summation and substitution
// order -- so every column of a batched solve has the bits of the solve done alone (tests/test_gpu_batch.py).
// The Krylov recurrences stay per system (their basis vectors are not shared: nothing to amortise); solver.hip drives K
// of them in lockstep around these kernels.
#include "nkp_dev.h"

#define BT_THREADS 256
#define BT_WAVES (BT_THREADS / NKP_WAVE)

static inline int bt_grid (int64_t n) { int64_t g = (n + BT_THREADS - 1) / BT_THREADS; return (int) (g < 1 ? 1 : g > 65535 * 16 ? 65535 * 16 : g); }

// ---------------------------------------------------------------- interleave / de-interleave
// X[i * K + k] = src_k[i]; systems without a vector (src_k == NULL) contribute zeros
struct BatchPtrs { const double *p[NKP_BATCH_MAX]; };
struct BatchOutPtrs { double *p[NKP_BATCH_MAX]; };

template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void interleave_kernel (BatchPtrs src, double *__restrict__ X, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; i < n; i += stride) {
#pragma unroll
      for (int k = 0; k < K; k++) X[i * K + k] = src.p[k] ? src.p[k][i] : 0.0;
   }
}

template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void deinterleave_kernel (const double *__restrict__ X, BatchOutPtrs dst, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; i < n; i += stride) {
#pragma unroll
      for (int k = 0; k < K; k++)
         if (dst.p[k]) dst.p[k][i] = X[i * K + k];
   }
}

void launch_interleave (int K, const double *const *src, double *X, int64_t n, hipStream_t st)
{
   BatchPtrs P;
   for (int k = 0; k < NKP_BATCH_MAX; k++) P.p[k] = k < K ? src[k] : nullptr;
   if (K == 2) hipLaunchKernelGGL (interleave_kernel<2>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, P, X, n);
   else if (K == 4) hipLaunchKernelGGL (interleave_kernel<4>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, P, X, n);
   else hipLaunchKernelGGL (interleave_kernel<8>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, P, X, n);
}

void launch_deinterleave (int K, const double *X, double *const *dst, int64_t n, hipStream_t st)
{
   BatchOutPtrs P;
   for (int k = 0; k < NKP_BATCH_MAX; k++) P.p[k] = k < K ? dst[k] : nullptr;
   if (K == 2) hipLaunchKernelGGL (deinterleave_kernel<2>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, X, P, n);
   else if (K == 4) hipLaunchKernelGGL (deinterleave_kernel<4>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, X, P, n);
   else hipLaunchKernelGGL (deinterleave_kernel<8>, dim3 (bt_grid (n)), dim3 (BT_THREADS), 0, st, X, P, n);
}

// ---------------------------------------------------------------- CSR SpMV, K columns
// csr_spmv_stream_kernel (spmv.hip) with the (value, column) stream of a row block read ONCE into registers and K / 2 passes
// over it: a pass gathers the 16-byte pair (x[c][2g], x[c][2g + 1]), parks both products in LDS and sums every row's segment
// in stored order.  MODE 0: y = A x   1: y = b - A x
template <int MODE, class VT, int K>
__global__ __launch_bounds__ (BT_THREADS)
void csr_spmv_batch_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk, int per_xcd, const int *__restrict__ rowptr,
                            const int *__restrict__ colind, const VT *__restrict__ val, const double *__restrict__ x,
                            double *__restrict__ y, const double *__restrict__ b)
{
   __shared__ double2 prod[NKP_SPMV_LDS_NNZ];
   __shared__ double2 wsum[BT_WAVES];
   constexpr int SLOTS = NKP_SPMV_LDS_NNZ / BT_THREADS;
   const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
   const int lb = xcd * per_xcd + idx;
   if (idx >= per_xcd || lb >= nrowblk) return;
   const int tid = threadIdx.x;
   const int *rowblk = rowblk_all + rb0;
   const int r0 = rowblk[lb], r1 = rowblk[lb + 1];
   const int e0 = rowptr[r0], e1 = rowptr[r1];
   const int cnt = e1 - e0;
   int seg0 = 0, seg1 = 0;
   if (r0 + tid < r1) { seg0 = rowptr[r0 + tid]; seg1 = rowptr[r0 + tid + 1]; }
   if (cnt > NKP_SPMV_LDS_NNZ) {
      // a single long row: strided accumulate + block reduction in the order of the single-vector kernel
      for (int g = 0; g < K / 2; g++) {
         double2 acc = make_double2 (0.0, 0.0);
         for (int e = e0 + tid; e < e1; e += BT_THREADS) {
            const double v = (double) val[e];
            const double2 xv = *reinterpret_cast<const double2 *> (x + (int64_t) colind[e] * K + 2 * g);
            acc.x += v * xv.x;
            acc.y += v * xv.y;
         }
         for (int off = NKP_WAVE / 2; off > 0; off >>= 1) { acc.x += __shfl_down (acc.x, off); acc.y += __shfl_down (acc.y, off); }
         if ((tid & (NKP_WAVE - 1)) == 0) wsum[tid / NKP_WAVE] = acc;
         __syncthreads ();
         if (tid == 0) {
            double2 s = make_double2 (0.0, 0.0);
            for (int w = 0; w < BT_WAVES; w++) { s.x += wsum[w].x; s.y += wsum[w].y; }
            if (MODE == 1) { const double2 bv = *reinterpret_cast<const double2 *> (b + (int64_t) r0 * K + 2 * g); s.x = bv.x - s.x; s.y = bv.y - s.y; }
            *reinterpret_cast<double2 *> (y + (int64_t) r0 * K + 2 * g) = s;
         }
         __syncthreads ();
      }
      return;
   }
   VT v[SLOTS];
   int c[SLOTS];
#pragma unroll
   for (int u = 0; u < SLOTS; u++) {
      const int k = tid + u * BT_THREADS;
      v[u] = k < cnt ? val[e0 + k] : (VT) 0;
      c[u] = k < cnt ? colind[e0 + k] : 0;
   }
   // all K values of a gathered row are requested together: the two 16-byte halves of a row of x sit in one cache line, so
   // the second load rides on the first one's L2 request (first version: one pass per pair = twice the L2 requests, and the
   // batched kernel was L2-request-bound at 1.9 TB/s)
   double2 xg[K / 2][SLOTS];
#pragma unroll
   for (int u = 0; u < SLOTS; u++) {
      const double *xr = x + (int64_t) c[u] * K;
#pragma unroll
      for (int g = 0; g < K / 2; g++) xg[g][u] = *reinterpret_cast<const double2 *> (xr + 2 * g);
   }
#pragma unroll
   for (int g = 0; g < K / 2; g++) {
#pragma unroll
      for (int u = 0; u < SLOTS; u++) {
         const int k = tid + u * BT_THREADS;
         if (k < cnt) prod[k] = make_double2 ((double) v[u] * xg[g][u].x, (double) v[u] * xg[g][u].y);
      }
      __syncthreads ();
      const int r = r0 + tid;
      if (r < r1) {
         const int s0 = seg0 - e0, s1 = seg1 - e0;
         double2 acc = make_double2 (0.0, 0.0);
#pragma unroll 4
         for (int k = s0; k < s1; k++) { acc.x += prod[k].x; acc.y += prod[k].y; }
         if (MODE == 1) { const double2 bv = *reinterpret_cast<const double2 *> (b + (int64_t) r * K + 2 * g); acc.x = bv.x - acc.x; acc.y = bv.y - acc.y; }
         *reinterpret_cast<double2 *> (y + (int64_t) r * K + 2 * g) = acc;
      }
      if (g + 1 < K / 2) __syncthreads ();
   }
}

// The other way round: the workgroup stages the row block's (value, column) stream in LDS with coalesced loads (8 bytes per
// entry instead of 16 K of products: 16 KB per workgroup, twice the workgroups per CU), then every row's lane walks its own
// segment, gathers the K-wide rows of x itself (GU entries in flight) and accumulates its K sums in registers -- same
// products, same stored order, one pass for any K.
// SPLIT: the K results of a row go to K separate vectors (split.p[k][r]; NULL = dropped) instead of the interleaved y -- the
// operator product at the end of an Arnoldi step writes every system's w directly.
template <int MODE, class VT, int K, bool SPLIT = false>
__global__ __launch_bounds__ (BT_THREADS)
void csr_spmv_batch_rows_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk, int per_xcd, const int *__restrict__ rowptr,
                                 const int *__restrict__ colind, const VT *__restrict__ val, const double *__restrict__ x,
                                 double *__restrict__ y, const double *__restrict__ b, BatchOutPtrs split = BatchOutPtrs ())
{
   __shared__ VT sv[NKP_SPMV_LDS_NNZ];
   __shared__ int sc[NKP_SPMV_LDS_NNZ];
   __shared__ double2 wsum[BT_WAVES];
   constexpr int SLOTS = NKP_SPMV_LDS_NNZ / BT_THREADS, GU = 16 / K;
   const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
   const int lb = xcd * per_xcd + idx;
   if (idx >= per_xcd || lb >= nrowblk) return;
   const int tid = threadIdx.x;
   const int *rowblk = rowblk_all + rb0;
   const int r0 = rowblk[lb], r1 = rowblk[lb + 1];
   const int e0 = rowptr[r0], e1 = rowptr[r1];
   const int cnt = e1 - e0;
   if (cnt > NKP_SPMV_LDS_NNZ) {
      for (int g = 0; g < K / 2; g++) {
         double2 acc = make_double2 (0.0, 0.0);
         for (int e = e0 + tid; e < e1; e += BT_THREADS) {
            const double v = (double) val[e];
            const double2 xv = *reinterpret_cast<const double2 *> (x + (int64_t) colind[e] * K + 2 * g);
            acc.x += v * xv.x;
            acc.y += v * xv.y;
         }
         for (int off = NKP_WAVE / 2; off > 0; off >>= 1) { acc.x += __shfl_down (acc.x, off); acc.y += __shfl_down (acc.y, off); }
         if ((tid & (NKP_WAVE - 1)) == 0) wsum[tid / NKP_WAVE] = acc;
         __syncthreads ();
         if (tid == 0) {
            double2 s = make_double2 (0.0, 0.0);
            for (int w = 0; w < BT_WAVES; w++) { s.x += wsum[w].x; s.y += wsum[w].y; }
            if (MODE == 1) { const double2 bv = *reinterpret_cast<const double2 *> (b + (int64_t) r0 * K + 2 * g); s.x = bv.x - s.x; s.y = bv.y - s.y; }
            if (SPLIT) { if (split.p[2 * g]) split.p[2 * g][r0] = s.x; if (split.p[2 * g + 1]) split.p[2 * g + 1][r0] = s.y; }
            else *reinterpret_cast<double2 *> (y + (int64_t) r0 * K + 2 * g) = s;
         }
         __syncthreads ();
      }
      return;
   }
   const int r = r0 + tid;
   int s0 = 0, s1 = 0;
   if (r < r1) { s0 = rowptr[r] - e0; s1 = rowptr[r + 1] - e0; }
   {
      VT v[SLOTS];
      int c[SLOTS];
#pragma unroll
      for (int u = 0; u < SLOTS; u++) {
         const int k = tid + u * BT_THREADS;
         v[u] = k < cnt ? val[e0 + k] : (VT) 0;
         c[u] = k < cnt ? colind[e0 + k] : 0;
      }
#pragma unroll
      for (int u = 0; u < SLOTS; u++) {
         const int k = tid + u * BT_THREADS;
         if (k < cnt) { sv[k] = v[u]; sc[k] = c[u]; }
      }
   }
   double2 bv[K / 2];
#pragma unroll
   for (int g = 0; g < K / 2; g++) bv[g] = (MODE == 1 && r < r1) ? *reinterpret_cast<const double2 *> (b + (int64_t) r * K + 2 * g) : make_double2 (0.0, 0.0);
   __syncthreads ();
   if (r >= r1) return;
   double2 acc[K / 2];
#pragma unroll
   for (int g = 0; g < K / 2; g++) acc[g] = make_double2 (0.0, 0.0);
   for (int k = s0; k < s1; k += GU) {
      double vv[GU];
      double2 xg[GU][K / 2];
#pragma unroll
      for (int u = 0; u < GU; u++) {
         const int kk = k + u < s1 ? k + u : s1 - 1;
         vv[u] = (double) sv[kk];
         const double *xr = x + (int64_t) sc[kk] * K;
#pragma unroll
         for (int g = 0; g < K / 2; g++) xg[u][g] = *reinterpret_cast<const double2 *> (xr + 2 * g);
      }
#pragma unroll
      for (int u = 0; u < GU; u++)
         if (k + u < s1) {
#pragma unroll
            for (int g = 0; g < K / 2; g++) { acc[g].x += vv[u] * xg[u][g].x; acc[g].y += vv[u] * xg[u][g].y; }
         }
   }
#pragma unroll
   for (int g = 0; g < K / 2; g++) {
      if (MODE == 1) { acc[g].x = bv[g].x - acc[g].x; acc[g].y = bv[g].y - acc[g].y; }
      if (SPLIT) { if (split.p[2 * g]) split.p[2 * g][r] = acc[g].x; if (split.p[2 * g + 1]) split.p[2 * g + 1][r] = acc[g].y; }
      else *reinterpret_cast<double2 *> (y + (int64_t) r * K + 2 * g) = acc[g];
   }
}

void launch_csr_spmv_batch (int K, const CsrDev &A, int rb0, int rb1, const double *x, double *y, const double *b, int mode, hipStream_t st)
{
   const int cnt = rb1 - rb0;
   if (cnt <= 0) return;
   const int per_xcd = (cnt + 7) / 8;
   const nkp_tuning &T = A.tune ? *A.tune : nkp_builtin_tuning ();
#define BSPMV(KERNEL_, MODE_, VT_, K_, VAL_) hipLaunchKernelGGL ((KERNEL_<MODE_, VT_, K_>), dim3 (per_xcd * 8), dim3 (BT_THREADS), 0, st, \
                                                                 A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, VAL_, x, y, b)
#define BSPMV_K(MODE_, VT_, VAL_) do { if (K == 8) BSPMV (csr_spmv_batch_rows_kernel, MODE_, VT_, 8, VAL_);      /* (the products-in-LDS variant stops at four) */ \
                                       else if (T.batch_spmv_rows) { if (K == 2) BSPMV (csr_spmv_batch_rows_kernel, MODE_, VT_, 2, VAL_); else BSPMV (csr_spmv_batch_rows_kernel, MODE_, VT_, 4, VAL_); } \
                                       else { if (K == 2) BSPMV (csr_spmv_batch_kernel, MODE_, VT_, 2, VAL_); else BSPMV (csr_spmv_batch_kernel, MODE_, VT_, 4, VAL_); } } while (0)
   if (A.valf) { if (mode == 0) BSPMV_K (0, float, A.valf); else BSPMV_K (1, float, A.valf); }
   else { if (mode == 0) BSPMV_K (0, double, A.val); else BSPMV_K (1, double, A.val); }
#undef BSPMV_K
#undef BSPMV
}

// y_k = A x_k for the K interleaved columns of x, every result in its own vector (dst[k] NULL: not wanted)
void launch_csr_spmv_batch_split (int K, const CsrDev &A, const double *x, double *const *dst, hipStream_t st)
{
   const int cnt = A.nrowblk;
   if (cnt <= 0) return;
   const int per_xcd = (cnt + 7) / 8;
   BatchOutPtrs P;
   for (int k = 0; k < NKP_BATCH_MAX; k++) P.p[k] = k < K ? dst[k] : nullptr;
#define BSPLIT(VT_, K_, VAL_) hipLaunchKernelGGL ((csr_spmv_batch_rows_kernel<0, VT_, K_, true>), dim3 (per_xcd * 8), dim3 (BT_THREADS), 0, st, \
                                                  A.rowblk, 0, cnt, per_xcd, A.rowptr, A.colind, VAL_, x, (double *) nullptr, (const double *) nullptr, P)
   if (A.valf) { if (K == 2) BSPLIT (float, 2, A.valf); else if (K == 4) BSPLIT (float, 4, A.valf); else BSPLIT (float, 8, A.valf); }
   else { if (K == 2) BSPLIT (double, 2, A.val); else if (K == 4) BSPLIT (double, 4, A.val); else BSPLIT (double, 8, A.val); }
#undef BSPLIT
}

// ---------------------------------------------------------------- grid transfer / permutation / coarsest solve, K columns
template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void restrict_sum_batch_kernel (const int *__restrict__ rptr, const int *__restrict__ ridx, const double *__restrict__ fine, double *__restrict__ coarse, int64_t nc)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t t = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; t < nc * K; t += stride) {
      const int64_t I = t / K;
      const int k = (int) (t % K);
      double acc = 0.0;
      for (int q = rptr[I]; q < rptr[I + 1]; q++) acc += fine[(int64_t) ridx[q] * K + k];
      coarse[t] = acc;
   }
}

template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void prolong_add_batch_kernel (const int *__restrict__ cmap, const double *__restrict__ coarse, double *__restrict__ fine, int64_t nf, double omega)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t t = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; t < nf * K; t += stride)
      fine[t] += omega * coarse[(int64_t) cmap[t / K] * K + (t % K)];
}

template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void gather_batch_kernel (const int *__restrict__ perm, const double *__restrict__ in, double *__restrict__ out, int64_t n, int scatter)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t t = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; t < n * K; t += stride) {
      const int64_t i = t / K;
      const int k = (int) (t % K);
      if (scatter) out[(int64_t) perm[i] * K + k] = in[t];
      else out[t] = in[(int64_t) perm[i] * K + k];
   }
}

// the entry and the exit of a batched cycle application, fused with the (de-)interleave of the per-system vectors:
//   entry: out[i * K + k] = src_k[perm[i]]        exit: z[perm[i] * K + k] = dst_k[perm[i]] = in[i * K + k]
template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void gather_interleave_kernel (const int *__restrict__ perm, BatchPtrs src, double *__restrict__ out, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; i < n; i += stride) {
      const int64_t pi = perm[i];
#pragma unroll
      for (int k = 0; k < K; k++) out[i * K + k] = src.p[k] ? src.p[k][pi] : 0.0;
   }
}

template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void scatter_split_kernel (const int *__restrict__ perm, const double *__restrict__ in, double *__restrict__ z, BatchOutPtrs dst, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * BT_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * BT_THREADS + threadIdx.x; i < n; i += stride) {
      const int64_t pi = perm[i];
#pragma unroll
      for (int k = 0; k < K; k++) {
         const double v = in[i * K + k];
         z[pi * K + k] = v;
         if (dst.p[k]) dst.p[k][pi] = v;
      }
   }
}

// one wave per output row, the K columns one after the other (each summed like dense_matvec_kernel sums its one)
template <int K>
__global__ __launch_bounds__ (BT_THREADS)
void dense_matvec_batch_kernel (const double *__restrict__ M, const double *__restrict__ x, double *__restrict__ y, int n)
{
   const int row = (int) ((blockIdx.x * BT_THREADS + threadIdx.x) / NKP_WAVE);
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   if (row >= n) return;
   const double *m = M + (int64_t) row * n;
   double acc[K];
#pragma unroll
   for (int k = 0; k < K; k++) acc[k] = 0.0;
   for (int c = lane; c < n; c += NKP_WAVE) {
      const double mv = m[c];
#pragma unroll
      for (int k = 0; k < K; k++) acc[k] += mv * x[(int64_t) c * K + k];
   }
#pragma unroll
   for (int k = 0; k < K; k++) {
      double v = acc[k];
      for (int off = NKP_WAVE / 2; off > 0; off >>= 1) v += __shfl_down (v, off);
      if (lane == 0) y[(int64_t) row * K + k] = v;
   }
}

#define BT_K(KERNEL, GRID, ...) do { if (K == 2) hipLaunchKernelGGL ((KERNEL<2>), GRID, dim3 (BT_THREADS), 0, st, __VA_ARGS__); \
                                     else if (K == 4) hipLaunchKernelGGL ((KERNEL<4>), GRID, dim3 (BT_THREADS), 0, st, __VA_ARGS__); \
                                     else hipLaunchKernelGGL ((KERNEL<8>), GRID, dim3 (BT_THREADS), 0, st, __VA_ARGS__); } while (0)

void launch_restrict_sum_batch (int K, const int *rptr, const int *ridx, const double *fine, double *coarse, int64_t nc, hipStream_t st)
{
   if (nc > 0) BT_K (restrict_sum_batch_kernel, dim3 (bt_grid (nc * K)), rptr, ridx, fine, coarse, nc);
}
void launch_prolong_add_batch (int K, const int *cmap, const double *coarse, double *fine, int64_t nf, double omega, hipStream_t st)
{
   if (nf > 0) BT_K (prolong_add_batch_kernel, dim3 (bt_grid (nf * K)), cmap, coarse, fine, nf, omega);
}
void launch_gather_batch (int K, const int *perm, const double *in, double *out, int64_t n, hipStream_t st)
{
   if (n > 0) BT_K (gather_batch_kernel, dim3 (bt_grid (n * K)), perm, in, out, n, 0);
}
void launch_scatter_batch (int K, const int *perm, const double *in, double *out, int64_t n, hipStream_t st)
{
   if (n > 0) BT_K (gather_batch_kernel, dim3 (bt_grid (n * K)), perm, in, out, n, 1);
}
void launch_gather_interleave (int K, const int *perm, const double *const *src, double *out, int64_t n, hipStream_t st)
{
   BatchPtrs P;
   for (int k = 0; k < NKP_BATCH_MAX; k++) P.p[k] = k < K ? src[k] : nullptr;
   if (n > 0) BT_K (gather_interleave_kernel, dim3 (bt_grid (n)), perm, P, out, n);
}
void launch_scatter_split (int K, const int *perm, const double *in, double *z, double *const *dst, int64_t n, hipStream_t st)
{
   BatchOutPtrs P;
   for (int k = 0; k < NKP_BATCH_MAX; k++) P.p[k] = k < K ? dst[k] : nullptr;
   if (n > 0) BT_K (scatter_split_kernel, dim3 (bt_grid (n)), perm, in, z, P, n);
}
void launch_dense_matvec_batch (int K, const double *Minv, const double *x, double *y, int n, hipStream_t st)
{
   if (n > 0) BT_K (dense_matvec_batch_kernel, dim3 ((n + BT_WAVES - 1) / BT_WAVES), Minv, x, y, n);
}

// ---------------------------------------------------------------- water columns, one per wave, K columns of right-hand sides
__device__ __forceinline__ double bt_readlane_f64 (double v, int lane)
{
   int lo = __double2loint (v), hi = __double2hiint (v);
   lo = __builtin_amdgcn_readlane (lo, lane);
   hi = __builtin_amdgcn_readlane (hi, lane);
   return __hiloint2double (hi, lo);
}

// band substitution of colblock_apply_kernel on K right-hand sides held in y[s][k] (lane = level, s = second register set of
// columns longer than a wave); identical operations per column, the K chains interleave in the pipeline
template <int P, int RPL, int K>
__device__ __forceinline__ void wave_band_solve (int len, int lane, double (&y)[RPL][K], const double (&invd)[RPL], const double (&L)[RPL][P], const double (&U)[RPL][P])
{
   for (int k = 0; k < len - 1; k++) {
      const int ks = k >> 6, kl = k & (NKP_WAVE - 1);
      double yk[K];
#pragma unroll
      for (int q = 0; q < K; q++) yk[q] = 0.0;
#pragma unroll
      for (int s = 0; s < RPL; s++)
         if (ks == s) {
#pragma unroll
            for (int q = 0; q < K; q++) yk[q] = bt_readlane_f64 (y[s][q], kl);
         }
#pragma unroll
      for (int s = 0; s < RPL; s++) {
         const int rel = s * NKP_WAVE + lane - k;
#pragma unroll
         for (int d = 1; d <= P; d++)
            if (rel == d) {
#pragma unroll
               for (int q = 0; q < K; q++) y[s][q] -= L[s][d - 1] * yk[q];
            }
      }
   }
   for (int k = len - 1; k >= 0; k--) {
      const int ks = k >> 6, kl = k & (NKP_WAVE - 1);
      double xk[K];
#pragma unroll
      for (int q = 0; q < K; q++) xk[q] = 0.0;
#pragma unroll
      for (int s = 0; s < RPL; s++)
         if (ks == s) {
#pragma unroll
            for (int q = 0; q < K; q++) {
               if (lane == kl) y[s][q] *= invd[s];
               xk[q] = bt_readlane_f64 (y[s][q], kl);
            }
         }
#pragma unroll
      for (int s = 0; s < RPL; s++) {
         const int rel = k - (s * NKP_WAVE + lane);
#pragma unroll
         for (int d = 1; d <= P; d++)
            if (rel == d) {
#pragma unroll
               for (int q = 0; q < K; q++) y[s][q] -= U[s][d - 1] * xk[q];
            }
      }
   }
}

template <int P, int RPL, bool R32>
__device__ __forceinline__ void wave_load_factors (int64_t n, int64_t r0, int len, int lane, const double *__restrict__ fac, double (&invd)[RPL], double (&L)[RPL][P], double (&U)[RPL][P])
{
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      invd[s] = 0.0;
#pragma unroll
      for (int q = 0; q < P; q++) { L[s][q] = 0.0; U[s][q] = 0.0; }
      if (li < len) {
         const int64_t r = r0 + li;
         invd[s] = fac[(int64_t) P * n + r];
         if (R32) invd[s] = (double) (float) invd[s];
#pragma unroll
         for (int q = 1; q <= P; q++) {
            L[s][q - 1] = fac[(int64_t) (P - q) * n + r];
            U[s][q - 1] = fac[(int64_t) (P + q) * n + r];
            if (R32) { L[s][q - 1] = (double) (float) L[s][q - 1]; U[s][q - 1] = (double) (float) U[s][q - 1]; }
         }
      }
   }
}

// z (+)= M^-1 rhs on the blocks [b_first, b_end), K columns
template <int P, int RPL, bool R32, int K>
__global__ __launch_bounds__ (BT_THREADS)
void colblock_apply_wave_batch_kernel (const int *__restrict__ blk_start, int b_first, int b_end, int64_t n, const double *__restrict__ fac,
                                       const double *__restrict__ rhs, double *__restrict__ z, int accumulate)
{
   const int blk = __builtin_amdgcn_readfirstlane ((int) ((blockIdx.x * BT_THREADS + threadIdx.x) / NKP_WAVE)) + b_first;
   if (blk >= b_end) return;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   const int r0 = blk_start[blk], len = blk_start[blk + 1] - r0;
   double y[RPL][K], invd[RPL], L[RPL][P], U[RPL][P];
   wave_load_factors<P, RPL, R32> (n, r0, len, lane, fac, invd, L, U);
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
#pragma unroll
      for (int q = 0; q < K; q++) y[s][q] = li < len ? rhs[((int64_t) r0 + li) * K + q] : 0.0;
   }
   wave_band_solve<P, RPL, K> (len, lane, y, invd, L, U);
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      if (li < len) {
#pragma unroll
         for (int q = 0; q < K; q++) {
            if (accumulate) z[((int64_t) r0 + li) * K + q] += y[s][q];
            else z[((int64_t) r0 + li) * K + q] = y[s][q];
         }
      }
   }
}

// gs_wave_kernel (colblock.hip) on K columns: residual of the column's rows, band solve, xout = x + z in one launch
#define BGS_UNROLL (K >= 8 ? 8 : 16)          // entries (x K values each) requested together
#define BGS_CAP 1536
template <int P, int RPL, class VT, bool R32, int K>
__global__ __launch_bounds__ (BT_THREADS)
void gs_wave_batch_kernel (const int *__restrict__ rowptr, const int *__restrict__ colind, const VT *__restrict__ val, const int *__restrict__ blk_start,
                           int b_first, int b_end, int64_t n, const double *__restrict__ fac, const double *__restrict__ xa, const double *__restrict__ xb,
                           int split, const double *__restrict__ b, double *__restrict__ xout, const int4 *__restrict__ desc)
{
   // the column's entries through LDS with coalesced loads, like gs_wave_kernel (colblock.hip)
   extern __shared__ unsigned char bgs_lds[];
   const int wv = threadIdx.x / NKP_WAVE;
   int *sc = reinterpret_cast<int *> (bgs_lds) + wv * BGS_CAP;
   VT *sv = reinterpret_cast<VT *> (bgs_lds + (size_t) BT_WAVES * BGS_CAP * sizeof (int)) + wv * BGS_CAP;
   const int blk = __builtin_amdgcn_readfirstlane ((int) ((blockIdx.x * BT_THREADS + threadIdx.x) / NKP_WAVE)) + b_first;
   const bool act = blk < b_end;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   int4 d4 = make_int4 (0, 0, 0, 0);
   if (act) {
      if (desc) d4 = desc[blk];
      else {
         d4.x = blk_start[blk];
         d4.y = blk_start[blk + 1] - d4.x;
         d4.z = d4.y > 0 ? rowptr[d4.x] : 0;
         d4.w = d4.y > 0 ? rowptr[d4.x + d4.y] - d4.z : 0;
      }
   }
   const int r0 = d4.x, len = d4.y;
   double y[RPL][K], xold[RPL][K], invd[RPL], L[RPL][P], U[RPL][P];
   int e0[RPL], rl[RPL];
   const int e_begin = d4.z, e_total = d4.w;
   const bool staged = e_total <= BGS_CAP;
   if (staged) {
      for (int k0 = 0; k0 < e_total; k0 += NKP_WAVE * 8) {
         int tc[8];
         VT tv[8];
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const int k = k0 + u * NKP_WAVE + lane;
            tc[u] = k < e_total ? colind[e_begin + k] : 0;
            tv[u] = k < e_total ? val[e_begin + k] : (VT) 0;
         }
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const int k = k0 + u * NKP_WAVE + lane;
            if (k < e_total) { sc[k] = tc[u]; sv[k] = tv[u]; }
         }
      }
   }
   wave_load_factors<P, RPL, R32> (n, r0, len, lane, fac, invd, L, U);
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      e0[s] = 0; rl[s] = 0;
#pragma unroll
      for (int q = 0; q < K; q++) { y[s][q] = 0.0; xold[s][q] = 0.0; }
      if (li < len) {
         const int64_t r = r0 + li;
         e0[s] = rowptr[r];
         rl[s] = rowptr[r + 1] - e0[s];
         const double *xo = (r < split) ? xa : xb;
#pragma unroll
         for (int q = 0; q < K; q++) { y[s][q] = b[r * K + q]; xold[s][q] = xo[r * K + q]; }
      }
   }
   __syncthreads ();
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      double acc[K];
#pragma unroll
      for (int q = 0; q < K; q++) acc[q] = 0.0;
      for (int k0 = 0; __any (k0 < rl[s]); k0 += BGS_UNROLL) {
         int cc[BGS_UNROLL];
         VT vv[BGS_UNROLL];
         if (staged) {
            const int off = e0[s] - e_begin + k0;
#pragma unroll
            for (int u = 0; u < BGS_UNROLL; u++) {
               const bool ok = k0 + u < rl[s];
               cc[u] = ok ? sc[off + u] : 0;
               vv[u] = ok ? sv[off + u] : (VT) 0;
            }
         } else {
#pragma unroll
            for (int u = 0; u < BGS_UNROLL; u++) {
               const bool ok = k0 + u < rl[s];
               cc[u] = ok ? colind[e0[s] + k0 + u] : 0;
               vv[u] = ok ? val[e0[s] + k0 + u] : (VT) 0;
            }
         }
         double2 xg[K / 2][BGS_UNROLL];
#pragma unroll
         for (int u = 0; u < BGS_UNROLL; u++) {
            const double *xs = ((cc[u] < split) ? xa : xb) + (int64_t) cc[u] * K;
#pragma unroll
            for (int h = 0; h < K / 2; h++) xg[h][u] = *reinterpret_cast<const double2 *> (xs + 2 * h);
         }
#pragma unroll
         for (int u = 0; u < BGS_UNROLL; u++)
            if (k0 + u < rl[s]) {
#pragma unroll
               for (int h = 0; h < K / 2; h++) { acc[2 * h] += (double) vv[u] * xg[h][u].x; acc[2 * h + 1] += (double) vv[u] * xg[h][u].y; }
            }
      }
      if (s * NKP_WAVE + lane < len) {
#pragma unroll
         for (int q = 0; q < K; q++) y[s][q] -= acc[q];
      }
   }
   wave_band_solve<P, RPL, K> (len, lane, y, invd, L, U);
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      if (li < len) {
#pragma unroll
         for (int q = 0; q < K; q++) xout[((int64_t) r0 + li) * K + q] = xold[s][q] + y[s][q];
      }
   }
}

static inline dim3 bt_wave_grid (int nblk) { return dim3 ((nblk + BT_WAVES - 1) / BT_WAVES); }

void launch_colblock_apply_wave_batch (int K, const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, int r32, hipStream_t st)
{
   if (b1 <= b0) return;
   const int rpl = B.max_len <= NKP_WAVE ? 1 : 2;
#define CW_GO(PP, RR, R32_, K_) hipLaunchKernelGGL ((colblock_apply_wave_batch_kernel<PP, RR, R32_, K_>), bt_wave_grid (b1 - b0), dim3 (BT_THREADS), 0, st, \
                                                     B.blk_start, b0, b1, B.n, B.fac, r, z, accumulate)
#define CW_K(PP, RR, R32_) do { if (K == 2) CW_GO (PP, RR, R32_, 2); else if (K == 4) CW_GO (PP, RR, R32_, 4); else CW_GO (PP, RR, R32_, 8); } while (0)
#define CW_R(PP, RR) do { if (r32) CW_K (PP, RR, true); else CW_K (PP, RR, false); } while (0)
   if (B.P == 1) { if (rpl == 1) CW_R (1, 1); else CW_R (1, 2); }
   else if (B.P == 2) { if (rpl == 1) CW_R (2, 1); else CW_R (2, 2); }
   else { if (rpl == 1) CW_R (4, 1); else CW_R (4, 2); }
#undef CW_R
#undef CW_K
#undef CW_GO
}

void launch_gs_wave_batch (int K, const CsrDev &L, const ColBlocksDev &B, int b0, int b1, const double *xa, const double *xb, int split, const double *b, double *xout,
                           int r32, hipStream_t st)
{
   if (b1 <= b0) return;
   const int rpl = B.max_len <= NKP_WAVE ? 1 : 2;
#define GW_GO(PP, RR, VT_, R32_, K_, VAL_) do { const size_t lds_ = (size_t) BT_WAVES * BGS_CAP * (sizeof (int) + sizeof (VT_));                                   \
                                                static bool opted_ = false;                                                                                             \
                                                if (lds_ > 48 * 1024 && !opted_) { (void) hipFuncSetAttribute ((const void *) gs_wave_batch_kernel<PP, RR, VT_, R32_, K_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_); opted_ = true; } \
                                                hipLaunchKernelGGL ((gs_wave_batch_kernel<PP, RR, VT_, R32_, K_>), bt_wave_grid (b1 - b0), dim3 (BT_THREADS), lds_, st, \
                                                                    L.rowptr, L.colind, VAL_, B.blk_start, b0, b1, B.n, B.fac, xa, xb, split, b, xout, reinterpret_cast<const int4 *> (B.wave_desc)); } while (0)
#define GW_K(PP, RR, VT_, R32_, VAL_) do { if (K == 2) GW_GO (PP, RR, VT_, R32_, 2, VAL_); else if (K == 4) GW_GO (PP, RR, VT_, R32_, 4, VAL_); else GW_GO (PP, RR, VT_, R32_, 8, VAL_); } while (0)
#define GW_PR(PP, RR) do { if (L.valf) { if (r32) GW_K (PP, RR, float, true, L.valf); else GW_K (PP, RR, float, false, L.valf); } \
                           else { if (r32) GW_K (PP, RR, double, true, L.val); else GW_K (PP, RR, double, false, L.val); } } while (0)
   if (B.P == 1) { if (rpl == 1) GW_PR (1, 1); else GW_PR (1, 2); }
   else if (B.P == 2) { if (rpl == 1) GW_PR (2, 1); else GW_PR (2, 2); }
   else { if (rpl == 1) GW_PR (4, 1); else GW_PR (4, 2); }
#undef GW_PR
#undef GW_K
#undef GW_GO
}
