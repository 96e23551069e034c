// Dense inverse of the coarsest operator on the device: blocked Gauss-Jordan on f64 matrix cores.
//
// A level of a few thousand rows costs the cycle ~200 us whatever its size -- 16 launches at their latency floor -- while
// a dense inverse applied as ONE matrix-vector product costs its bytes: n^2 x 4 at f32, 35 us for the 7177 rows of the 1
// degree hierarchy's fifth level.  So the hierarchy now stops at <= 8000 rows instead of <= 3000 (nkp_tuning.ml_coarsest_rows),
// which needs an inverse of that size inside the setup budget: 2 n^3 = 7.4e11 flops.  The unblocked elimination of
// multilevel.hip (four launches and two sweeps over both matrices per pivot) moves 32 n^3 bytes -- 3.1 s at n = 7177.
//
// Here: in-place block Gauss-Jordan, 64 x 64 blocks, no pivoting across blocks (the operator is the Galerkin product of an
// M-matrix: elimination without pivoting is stable; a pivot below 1e-14 of the largest diagonal entry raises a flag and the
// caller falls back to the pivoted routine).  Step k:
//    D      = inverse of the diagonal block                       (one workgroup, in LDS)
//    P      = column panel k (copied out), column panel k := 0    \ one launch
//    row panel k := D x row panel k, its diagonal block := D      /
//    every other block row I: A[I][:] -= P[I] x row panel k       (64 x 64 x 64 products on v_mfma_f64_16x16x4f64)
// 2 n^3 flops in the last kernel, 16 n^2 bytes per step: 113 steps of ~0.3 ms at n = 7177.
#include "nkp_dev.h"

#include <math.h>
#include <vector>

#define DNB 64
#define DLD 65            // LDS leading dimension of the diagonal block
typedef double v4d __attribute__ ((ext_vector_type (4)));

__global__ __launch_bounds__ (256)
void dense_scatter_kernel (const int *__restrict__ rowptr, const int *__restrict__ col, const double *__restrict__ val, int n, int np, double *__restrict__ a)
{
   const int r = blockIdx.x * 256 + threadIdx.x;
   if (r >= np) return;
   if (r >= n) { a[(size_t) r * np + r] = 1.0; return; }         // identity padding up to a multiple of the block size
   for (int e = rowptr[r]; e < rowptr[r + 1]; e++) a[(size_t) r * np + col[e]] = val[e];
}

__global__ __launch_bounds__ (256)
void bgj_diag_kernel (const double *__restrict__ a, int np, int k, double *__restrict__ D, int *__restrict__ flag, double tiny)
{
   __shared__ double s[DNB * DLD];
   __shared__ double fcol[DNB], prow[DNB];
   const int tid = threadIdx.x;
   const double *t0 = a + ((size_t) k * DNB) * np + (size_t) k * DNB;
   for (int idx = tid; idx < DNB * DNB; idx += 256) s[(idx >> 6) * DLD + (idx & 63)] = t0[(size_t) (idx >> 6) * np + (idx & 63)];
   __syncthreads ();
   for (int p = 0; p < DNB; p++) {
      double piv = s[p * DLD + p];
      if (!(fabs (piv) > tiny)) { if (tid == 0) flag[0] = 1; piv = 1.0; }
      const double d = 1.0 / piv;
      if (tid < DNB) { fcol[tid] = s[tid * DLD + p]; prow[tid] = (tid == p) ? d : s[p * DLD + tid] * d; }
      __syncthreads ();
      for (int idx = tid; idx < DNB * DNB; idx += 256) {
         const int r = idx >> 6, c = idx & 63;
         s[r * DLD + c] = (r == p) ? prow[c] : (c == p) ? -fcol[r] * d : s[r * DLD + c] - fcol[r] * prow[c];
      }
      __syncthreads ();
   }
   for (int idx = tid; idx < DNB * DNB; idx += 256) D[idx] = s[(idx >> 6) * DLD + (idx & 63)];
}

// acc (this wave's 32 x 32 quadrant of a 64 x 64 tile) += sign * Ag (64 x 64, leading dimension lda) x Bg (64 x 64, ldb), staged
// through LDS in two halves of 32 in k (33 KB per workgroup).
// v_mfma_f64_16x16x4f64: lane l feeds A[l % 16][l / 16] and B[l / 16][l % 16]; accumulator element v is C[l / 16 + 4 v][l % 16]
// (registers step by FOUR rows -- unlike the f32 16x16x4 instruction, whose registers hold four consecutive rows)
#define DKH 32
__device__ __forceinline__ void bgj_tile_mma (const double *__restrict__ Ag, size_t lda, double sign, const double *__restrict__ Bg, size_t ldb,
                                              double *sA /* [64][DKH + 1] */, double *sB /* [DKH][64] */, v4d (&acc)[2][2], int tid)
{
   const int wave = tid >> 6, lane = tid & 63, wr = wave >> 1, wc = wave & 1;
   const int l16 = lane & 15, lk = lane >> 4;
   for (int h = 0; h < DNB / DKH; h++) {
      __syncthreads ();
      for (int idx = tid; idx < DNB * DKH; idx += 256) {
         const int r = idx / DKH, c = idx % DKH;            // A: 64 rows x 32 columns of this half
         sA[r * (DKH + 1) + c] = sign * Ag[(size_t) r * lda + h * DKH + c];
         const int rb = idx >> 6, cb = idx & 63;            // B: 32 rows x 64 columns
         sB[rb * DNB + cb] = Bg[(size_t) (h * DKH + rb) * ldb + cb];
      }
      __syncthreads ();
#pragma unroll 4
      for (int k0 = 0; k0 < DKH; k0 += 4) {
         double av[2], bv[2];
#pragma unroll
         for (int m = 0; m < 2; m++) av[m] = sA[(wr * 32 + m * 16 + l16) * (DKH + 1) + k0 + lk];
#pragma unroll
         for (int q = 0; q < 2; q++) bv[q] = sB[(k0 + lk) * DNB + wc * 32 + q * 16 + l16];
#pragma unroll
         for (int m = 0; m < 2; m++)
#pragma unroll
            for (int q = 0; q < 2; q++) acc[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64 (av[m], bv[q], acc[m][q], 0, 0, 0);
      }
   }
}

// the accumulator's elements in a row-major tile (leading dimension ld)
#define BGJ_FOR_ACC(EXPR)                                                                                                   \
   _Pragma ("unroll") for (int m = 0; m < 2; m++)                                                                           \
   _Pragma ("unroll") for (int q = 0; q < 2; q++)                                                                           \
   _Pragma ("unroll") for (int v = 0; v < 4; v++) {                                                                         \
      const size_t off = (size_t) ((tid >> 7) * 32 + m * 16 + ((tid & 63) >> 4) + 4 * v) * ld + (((tid >> 6) & 1) * 32 + q * 16 + (tid & 15)); \
      EXPR;                                                                                                                 \
   }

// grid (np / 64, 2).  y = 0: block row t of the column panel -> P, then zeroed.  y = 1: block column t of the row panel := D x itself
__global__ __launch_bounds__ (256)
void bgj_panels_kernel (double *__restrict__ a, int np, int k, const double *__restrict__ D, double *__restrict__ P)
{
   __shared__ double sA[DNB * (DKH + 1)], sB[DKH * DNB];
   const int tid = threadIdx.x, t = blockIdx.x;
   const size_t ld = (size_t) np;
   if (blockIdx.y == 0) {
      if (t == k) return;
      double *src = a + ((size_t) t * DNB) * ld + (size_t) k * DNB;
      for (int idx = tid; idx < DNB * DNB; idx += 256) {
         const int r = idx >> 6, c = idx & 63;
         P[((size_t) t * DNB + r) * DNB + c] = src[(size_t) r * ld + c];
         src[(size_t) r * ld + c] = 0.0;
      }
      return;
   }
   double *dst = a + ((size_t) k * DNB) * ld + (size_t) t * DNB;
   if (t == k) {
      for (int idx = tid; idx < DNB * DNB; idx += 256) dst[(size_t) (idx >> 6) * ld + (idx & 63)] = D[idx];
      return;
   }
   v4d acc[2][2];
#pragma unroll
   for (int m = 0; m < 2; m++)
#pragma unroll
      for (int q = 0; q < 2; q++) acc[m][q] = (v4d) { 0.0, 0.0, 0.0, 0.0 };
   bgj_tile_mma (D, DNB, 1.0, dst, ld, sA, sB, acc, tid);
   __syncthreads ();                                     // every wave has read the old tile (through LDS) before anyone overwrites it
   BGJ_FOR_ACC (dst[off] = acc[m][q][v])
}

// grid (np / 64, np / 64): block (I = y, J = x), I != k: A[I][J] -= P[I] x row panel k [J]
__global__ __launch_bounds__ (256)
void bgj_update_kernel (double *__restrict__ a, int np, int k, const double *__restrict__ P)
{
   __shared__ double sA[DNB * (DKH + 1)], sB[DKH * DNB];
   const int I = blockIdx.y, J = blockIdx.x, tid = threadIdx.x;
   if (I == k) return;
   const size_t ld = (size_t) np;
   double *ct = a + ((size_t) I * DNB) * ld + (size_t) J * DNB;
   v4d acc[2][2];
   BGJ_FOR_ACC (acc[m][q][v] = ct[off])
   bgj_tile_mma (P + (size_t) I * DNB * DNB, DNB, -1.0, a + ((size_t) k * DNB) * ld + (size_t) J * DNB, ld, sA, sB, acc, tid);
   BGJ_FOR_ACC (ct[off] = acc[m][q][v])
}

// the n x n corner of the padded result: compact f64 (row-major, stride n) and, if wanted, f32 with rows padded to ldf
__global__ __launch_bounds__ (256)
void dense_export_kernel (const double *__restrict__ a, int n, int np, double *__restrict__ out, float *__restrict__ outf, int ldf)
{
   const int r = blockIdx.y;
   const int c = blockIdx.x * 256 + threadIdx.x;
   if (c < n) out[(size_t) r * n + c] = a[(size_t) r * np + c];
   if (outf && c < ldf) outf[(size_t) r * ldf + c] = c < n ? (float) a[(size_t) r * np + c] : 0.0f;
}

// CSR (host) of an n x n operator -> its inverse on the device: *inv_out n x n f64 row-major; *invf_out (if not NULL) the f32
// copy with leading dimension *ldf_out.  Returns 0, 1 = a pivot too small for elimination without pivoting (nothing
// allocated: take the pivoted routine), -1 = out of device memory / HIP error
int dense_inverse_blocked_device (int n, const int *h_rowptr, const int *h_col, const double *h_val, double **inv_out, float **invf_out, int *ldf_out,
                                  size_t *bytes, hipStream_t st)
{
   const int nb = (n + DNB - 1) / DNB, np = nb * DNB;
   const size_t nnz = (size_t) h_rowptr[n];
   double maxdiag = 0.0;
   for (int r = 0; r < n; r++)
      for (int e = h_rowptr[r]; e < h_rowptr[r + 1]; e++)
         if (h_col[e] == r && fabs (h_val[e]) > maxdiag) maxdiag = fabs (h_val[e]);
   double *a = nullptr, *D = nullptr, *P = nullptr, *val = nullptr, *out = nullptr;
   float *outf = nullptr;
   int *rowptr = nullptr, *col = nullptr, *flag = nullptr;
   const int ldf = (n + 3) & ~3;
   bool ok = hipMalloc ((void **) &a, (size_t) np * np * sizeof (double)) == hipSuccess && hipMalloc ((void **) &D, DNB * DNB * sizeof (double)) == hipSuccess &&
             hipMalloc ((void **) &P, (size_t) np * DNB * sizeof (double)) == hipSuccess && hipMalloc ((void **) &val, (nnz ? nnz : 1) * sizeof (double)) == hipSuccess &&
             hipMalloc ((void **) &rowptr, (size_t) (n + 1) * sizeof (int)) == hipSuccess && hipMalloc ((void **) &col, (nnz ? nnz : 1) * sizeof (int)) == hipSuccess &&
             hipMalloc ((void **) &flag, sizeof (int)) == hipSuccess;
   ok = ok && hipMemcpyAsync (rowptr, h_rowptr, (size_t) (n + 1) * sizeof (int), hipMemcpyHostToDevice, st) == hipSuccess &&
        hipMemcpyAsync (col, h_col, nnz * sizeof (int), hipMemcpyHostToDevice, st) == hipSuccess &&
        hipMemcpyAsync (val, h_val, nnz * sizeof (double), hipMemcpyHostToDevice, st) == hipSuccess &&
        hipMemsetAsync (a, 0, (size_t) np * np * sizeof (double), st) == hipSuccess && hipMemsetAsync (flag, 0, sizeof (int), st) == hipSuccess;
   int rc = ok ? 0 : -1;
   if (ok) {
      hipLaunchKernelGGL (dense_scatter_kernel, dim3 ((np + 255) / 256), dim3 (256), 0, st, rowptr, col, val, n, np, a);
      for (int k = 0; k < nb; k++) {
         hipLaunchKernelGGL (bgj_diag_kernel, dim3 (1), dim3 (256), 0, st, a, np, k, D, flag, 1e-14 * maxdiag);
         hipLaunchKernelGGL (bgj_panels_kernel, dim3 (nb, 2), dim3 (256), 0, st, a, np, k, D, P);
         hipLaunchKernelGGL (bgj_update_kernel, dim3 (nb, nb), dim3 (256), 0, st, a, np, k, P);
      }
      int hflag = 0;
      if (hipMemcpyAsync (&hflag, flag, sizeof (int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize (st) != hipSuccess) rc = -1;
      else if (hflag) rc = 1;
   }
   if (rc == 0) {
      ok = hipMalloc ((void **) &out, (size_t) n * n * sizeof (double)) == hipSuccess && (!invf_out || hipMalloc ((void **) &outf, (size_t) n * ldf * sizeof (float)) == hipSuccess);
      if (!ok) rc = -1;
      else {
         hipLaunchKernelGGL (dense_export_kernel, dim3 ((np + 255) / 256, n), dim3 (256), 0, st, a, n, np, out, outf, ldf);
         if (hipStreamSynchronize (st) != hipSuccess) rc = -1;
      }
   }
   for (void *q : { (void *) a, (void *) D, (void *) P, (void *) val, (void *) rowptr, (void *) col, (void *) flag })
      if (q) (void) hipFree (q);
   if (rc != 0) {
      if (out) (void) hipFree (out);
      if (outf) (void) hipFree (outf);
      return rc;
   }
   *inv_out = out;
   *bytes += (size_t) n * n * sizeof (double);
   if (invf_out) { *invf_out = outf; *ldf_out = ldf; *bytes += (size_t) n * ldf * sizeof (float); }
   return 0;
}

// ---------------------------------------------------------------- y = M x with the f32 copy: one wave per row, 16-byte loads
// Summation order (fixed, the tail kernel of mltail.hip repeats it): lane l owns the column quads l, l + 64, ...; a quad adds its
// four products in column order; then the 64 partial sums go through the shuffle tree.  DMV_UNROLL quads are requested before the
// first is used (the first version had one load in flight per lane: 74 us for the 206 MB of the 1 degree level, 2.8 TB/s).
#define DMV_UNROLL 4
#define DMV_ROWS 4
// ... and a wave takes DMV_ROWS rows at once: the loads of x (32 bytes per quad and lane, from L1/L2 -- x is 57 KB) were two thirds
// of the L1 requests of the one-row kernel; four rows share them.  Measured for the 206 MB of the 1 degree level (rows x quads in
// flight per lane): 1 x 1 75 us, 1 x 8 80, 4 x 4 **64**, 2 x 8 75, 8 x 2 80, 4 x 8 117 (256 VGPRs), x staged in LDS by 16-wave workgroups 69 --
// the matrix stream itself does not get past 3.2 TB/s in this one-shot launch; 4 x 4 stays.
__global__ __launch_bounds__ (256)
void dense_matvec_f32_kernel (const float *__restrict__ M, int ld, const double *__restrict__ x, double *__restrict__ y, int n)
{
   const int row0 = (int) ((blockIdx.x * 256 + threadIdx.x) / NKP_WAVE) * DMV_ROWS;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   if (row0 >= n) return;
   const float4 *m[DMV_ROWS];
#pragma unroll
   for (int r = 0; r < DMV_ROWS; r++) m[r] = reinterpret_cast<const float4 *> (M + (size_t) (row0 + r < n ? row0 + r : n - 1) * ld);
   const int nq = (n + 3) >> 2;                       // quads of a row (ld is a multiple of 4, the tail of the last quad is zero)
   double acc[DMV_ROWS];
#pragma unroll
   for (int r = 0; r < DMV_ROWS; r++) acc[r] = 0.0;
   for (int q0 = lane; q0 < nq; q0 += NKP_WAVE * DMV_UNROLL) {
      float4 v[DMV_ROWS][DMV_UNROLL];
#pragma unroll
      for (int u = 0; u < DMV_UNROLL; u++) {
         const int q = q0 + u * NKP_WAVE;
#pragma unroll
         for (int r = 0; r < DMV_ROWS; r++) v[r][u] = q < nq ? m[r][q] : make_float4 (0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int u = 0; u < DMV_UNROLL; u++) {
         const int c = (q0 + u * NKP_WAVE) * 4;
         const double x0 = c < n ? x[c] : 0.0, x1 = c + 1 < n ? x[c + 1] : 0.0, x2 = c + 2 < n ? x[c + 2] : 0.0, x3 = c + 3 < n ? x[c + 3] : 0.0;
#pragma unroll
         for (int r = 0; r < DMV_ROWS; r++) {
            if (c < n) acc[r] += (double) v[r][u].x * x0;
            if (c + 1 < n) acc[r] += (double) v[r][u].y * x1;
            if (c + 2 < n) acc[r] += (double) v[r][u].z * x2;
            if (c + 3 < n) acc[r] += (double) v[r][u].w * x3;
         }
      }
   }
#pragma unroll
   for (int r = 0; r < DMV_ROWS; r++) {
      double a = acc[r];
      for (int off = NKP_WAVE / 2; off > 0; off >>= 1) a += __shfl_down (a, off);
      if (lane == 0 && row0 + r < n) y[row0 + r] = a;
   }
}

void launch_dense_matvec_f32 (const float *Minv, int ld, const double *x, double *y, int n, hipStream_t st)
{
   const int waves = (n + DMV_ROWS - 1) / DMV_ROWS;
   if (n > 0) hipLaunchKernelGGL (dense_matvec_f32_kernel, dim3 ((waves + 3) / 4), dim3 (256), 0, st, Minv, ld, x, y, n);
}

// K interleaved right-hand sides; every (row, column k) accumulates exactly like the single-vector kernel above.  A wave takes
// BR rows: the K-wide rows of x (32 K bytes per quad and lane) are loaded once for all of them (the one-row version spent
// 216 us on the 1 degree level with K = 4, as much as four single products).
template <int K>
__global__ __launch_bounds__ (256)
void dense_matvec_f32_batch_kernel (const float *__restrict__ M, int ld, const double *__restrict__ x, double *__restrict__ y, int n)
{
   constexpr int BR = K >= 8 ? 2 : 4;
   const int row0 = (int) ((blockIdx.x * 256 + threadIdx.x) / NKP_WAVE) * BR;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   if (row0 >= n) return;
   const float4 *m[BR];
#pragma unroll
   for (int r = 0; r < BR; r++) m[r] = reinterpret_cast<const float4 *> (M + (size_t) (row0 + r < n ? row0 + r : n - 1) * ld);
   const int nq = (n + 3) >> 2;
   double acc[BR][K];
#pragma unroll
   for (int r = 0; r < BR; r++)
#pragma unroll
      for (int k = 0; k < K; k++) acc[r][k] = 0.0;
   for (int q = lane; q < nq; q += NKP_WAVE) {
      float4 v[BR];
#pragma unroll
      for (int r = 0; r < BR; r++) v[r] = m[r][q];
      const int c0 = q * 4;
      double xv[4][K];
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
         for (int k = 0; k < K; k += 2) {
            const double2 t = c0 + j < n ? *reinterpret_cast<const double2 *> (x + (size_t) (c0 + j) * K + k) : make_double2 (0.0, 0.0);
            xv[j][k] = t.x;
            xv[j][k + 1] = t.y;
         }
#pragma unroll
      for (int r = 0; r < BR; r++) {
         const float mv[4] = { v[r].x, v[r].y, v[r].z, v[r].w };
#pragma unroll
         for (int j = 0; j < 4; j++)
            if (c0 + j < n) {
#pragma unroll
               for (int k = 0; k < K; k++) acc[r][k] += (double) mv[j] * xv[j][k];
            }
      }
   }
#pragma unroll
   for (int r = 0; r < BR; r++)
#pragma unroll
      for (int k = 0; k < K; k++) {
         double a = acc[r][k];
         for (int off = NKP_WAVE / 2; off > 0; off >>= 1) a += __shfl_down (a, off);
         if (lane == 0 && row0 + r < n) y[(size_t) (row0 + r) * K + k] = a;
      }
}

void launch_dense_matvec_f32_batch (int K, const float *Minv, int ld, const double *x, double *y, int n, hipStream_t st)
{
   if (n <= 0) return;
   const int br = K >= 8 ? 2 : 4, waves = (n + br - 1) / br;
   const dim3 grid ((waves + 3) / 4);
   if (K == 2) hipLaunchKernelGGL ((dense_matvec_f32_batch_kernel<2>), grid, dim3 (256), 0, st, Minv, ld, x, y, n);
   else if (K == 4) hipLaunchKernelGGL ((dense_matvec_f32_batch_kernel<4>), grid, dim3 (256), 0, st, Minv, ld, x, y, n);
   else hipLaunchKernelGGL ((dense_matvec_f32_batch_kernel<8>), grid, dim3 (256), 0, st, Minv, ld, x, y, n);
}
