// Vector kernels of the Krylov drivers (gfx950).  All are HBM-bound streams: 16-byte loads
// per lane, grid-stride, 64-lane __shfl_down wave reductions, and a FIXED number of partial
// sums that a second tiny kernel adds in a fixed order -- so every inner product is bitwise
// reproducible (no float atomics) and, in the multi-GPU build, is the local contribution fed
// to one RCCL allreduce per Gram-Schmidt sweep.
//
// Stand in for the norm / update bookkeeping of pdgsrfs_ABXglobal (reference
// src/SuperLU_brief_tree.txt:20).  The fused multi-dot reads w once per chunk of 8 basis
// vectors and every basis vector exactly once.
#include "nkp_dev.h"

#include <stdint.h>

#define B1_THREADS 256

__device__ __forceinline__ double wave_sum (double v)
{
#pragma unroll
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) v += __shfl_down (v, off);
   return v;
}

// block-wide sum, result valid in thread 0
__device__ __forceinline__ double block_sum (double v, double *sh /* [B1_THREADS/64] */)
{
   v = wave_sum (v);
   const int lane = threadIdx.x & (NKP_WAVE - 1), wv = threadIdx.x / NKP_WAVE;
   __syncthreads ();
   if (lane == 0) sh[wv] = v;
   __syncthreads ();
   double s = 0.0;
   if (threadIdx.x == 0) {
#pragma unroll
      for (int w = 0; w < B1_THREADS / NKP_WAVE; w++) s += sh[w];
   }
   return s;
}

static inline int red_grid (int64_t n)
{
   int64_t g = (n + 2 * B1_THREADS - 1) / (2 * B1_THREADS);
   if (g < 1) g = 1;
   if (g > NKP_RED_BLOCKS) g = NKP_RED_BLOCKS;
   return (int) g;
}

// ---------------------------------------------------------------- multi-dot
// grid (nblk, nchunk).  partial[(chunk*nblk + blk)*(CHUNK+1) + c]; slot CHUNK of chunk 0 = w.w
template <class VT, class VT2>
__global__ __launch_bounds__ (B1_THREADS)
void multi_dot_kernel (const VT *__restrict__ V, int64_t ld, int k, const double *__restrict__ w,
                       int64_t n, double *__restrict__ partial)
{
   __shared__ double sh[B1_THREADS / NKP_WAVE];
   const int chunk = blockIdx.y;
   const int j0 = chunk * NKP_DOT_CHUNK;
   const int kc = (k - j0) < NKP_DOT_CHUNK ? (k - j0) : NKP_DOT_CHUNK;
   const VT *Vc = V + (int64_t) j0 * ld;
   double acc[NKP_DOT_CHUNK];
#pragma unroll
   for (int c = 0; c < NKP_DOT_CHUNK; c++) acc[c] = 0.0;
   double accw = 0.0;
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS * 2;
   for (int64_t i = ((int64_t) blockIdx.x * B1_THREADS + threadIdx.x) * 2; i < n; i += stride) {
      if (i + 1 < n) {
         const double2 w2 = *reinterpret_cast<const double2 *> (w + i);
         accw += w2.x * w2.x + w2.y * w2.y;
#pragma unroll
         for (int c = 0; c < NKP_DOT_CHUNK; c++)
            if (c < kc) {
               const VT2 v2 = *reinterpret_cast<const VT2 *> (Vc + (int64_t) c * ld + i);
               acc[c] += (double) v2.x * w2.x + (double) v2.y * w2.y;
            }
      } else {
         const double w1 = w[i];
         accw += w1 * w1;
#pragma unroll
         for (int c = 0; c < NKP_DOT_CHUNK; c++)
            if (c < kc) acc[c] += (double) Vc[(int64_t) c * ld + i] * w1;
      }
   }
   double *out = partial + ((int64_t) chunk * gridDim.x + blockIdx.x) * (NKP_DOT_CHUNK + 1);
#pragma unroll
   for (int c = 0; c < NKP_DOT_CHUNK; c++) {
      const double s = block_sum (acc[c], sh);
      if (threadIdx.x == 0) out[c] = s;
   }
   const double sw = block_sum (accw, sh);
   if (threadIdx.x == 0) out[NKP_DOT_CHUNK] = sw;
}

// one wave per output: out[j] = sum_blk partial[...], fixed order.  out[k] = w.w (chunk 0)
__global__ __launch_bounds__ (NKP_WAVE)
void multi_dot_finish_kernel (const double *__restrict__ partial, int nblk, int k, double *__restrict__ out)
{
   const int j = blockIdx.x;                   // 0..k  (k = the w.w slot)
   const int chunk = (j < k) ? j / NKP_DOT_CHUNK : 0;
   const int c = (j < k) ? j % NKP_DOT_CHUNK : NKP_DOT_CHUNK;
   const double *p = partial + (int64_t) chunk * nblk * (NKP_DOT_CHUNK + 1) + c;
   double s = 0.0;
   for (int b = threadIdx.x; b < nblk; b += NKP_WAVE) s += p[(int64_t) b * (NKP_DOT_CHUNK + 1)];
   s = wave_sum (s);
   if (threadIdx.x == 0) out[j] = s;
}

void launch_multi_dot (const void *V, int v_f32, int64_t ld, int k, const double *w, int64_t n, double *partial, double *out, hipStream_t st)
{
   const int g = red_grid (n);
   const int nchunk = k > 0 ? (k + NKP_DOT_CHUNK - 1) / NKP_DOT_CHUNK : 1;
   if (v_f32) hipLaunchKernelGGL ((multi_dot_kernel<float, float2>), dim3 (g, nchunk), dim3 (B1_THREADS), 0, st, (const float *) V, ld, k, w, n, partial);
   else hipLaunchKernelGGL ((multi_dot_kernel<double, double2>), dim3 (g, nchunk), dim3 (B1_THREADS), 0, st, (const double *) V, ld, k, w, n, partial);
   hipLaunchKernelGGL (multi_dot_finish_kernel, dim3 (k + 1), dim3 (NKP_WAVE), 0, st, partial, g, k, out);
}

// ---------------------------------------------------------------- w -= V h, with ||w||^2
#define B1_MAX_K NKP_MAX_K
template <class VT, class VT2>
__global__ __launch_bounds__ (B1_THREADS)
void update_w_kernel (const VT *__restrict__ V, int64_t ld, int k, const double *__restrict__ h,
                      double *__restrict__ w, int64_t n, double *__restrict__ partial, double sign)
{
   __shared__ double hs[B1_MAX_K];
   __shared__ double sh[B1_THREADS / NKP_WAVE];
   for (int j = threadIdx.x; j < k; j += B1_THREADS) hs[j] = sign * h[j];
   __syncthreads ();
   double nrm = 0.0;
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS * 2;
   for (int64_t i = ((int64_t) blockIdx.x * B1_THREADS + threadIdx.x) * 2; i < n; i += stride) {
      if (i + 1 < n) {
         double2 a = *reinterpret_cast<const double2 *> (w + i);
         int j = 0;
         for (; j + 4 <= k; j += 4) {
            const VT2 v0 = *reinterpret_cast<const VT2 *> (V + (int64_t) (j + 0) * ld + i);
            const VT2 v1 = *reinterpret_cast<const VT2 *> (V + (int64_t) (j + 1) * ld + i);
            const VT2 v2 = *reinterpret_cast<const VT2 *> (V + (int64_t) (j + 2) * ld + i);
            const VT2 v3 = *reinterpret_cast<const VT2 *> (V + (int64_t) (j + 3) * ld + i);
            a.x += hs[j] * v0.x; a.y += hs[j] * v0.y;
            a.x += hs[j + 1] * v1.x; a.y += hs[j + 1] * v1.y;
            a.x += hs[j + 2] * v2.x; a.y += hs[j + 2] * v2.y;
            a.x += hs[j + 3] * v3.x; a.y += hs[j + 3] * v3.y;
         }
         for (; j < k; j++) {
            const VT2 v0 = *reinterpret_cast<const VT2 *> (V + (int64_t) j * ld + i);
            a.x += hs[j] * v0.x; a.y += hs[j] * v0.y;
         }
         *reinterpret_cast<double2 *> (w + i) = a;
         nrm += a.x * a.x + a.y * a.y;
      } else {
         double a = w[i];
         for (int j = 0; j < k; j++) a += hs[j] * V[(int64_t) j * ld + i];
         w[i] = a;
         nrm += a * a;
      }
   }
   if (partial) {
      const double s = block_sum (nrm, sh);
      if (threadIdx.x == 0) partial[blockIdx.x] = s;
   }
}

__global__ __launch_bounds__ (NKP_WAVE)
void sum_partials_kernel (const double *__restrict__ partial, int nblk, double *__restrict__ out)
{
   double s = 0.0;
   for (int b = threadIdx.x; b < nblk; b += NKP_WAVE) s += partial[b];
   s = wave_sum (s);
   if (threadIdx.x == 0) out[0] = s;
}

void launch_update_w (const void *V, int v_f32, int64_t ld, int k, const double *h, double *w, int64_t n, double *partial, double *out_nrm2, hipStream_t st)
{
   const int g = red_grid (n);
   if (v_f32) hipLaunchKernelGGL ((update_w_kernel<float, float2>), dim3 (g), dim3 (B1_THREADS), 0, st, (const float *) V, ld, k, h, w, n, partial, -1.0);
   else hipLaunchKernelGGL ((update_w_kernel<double, double2>), dim3 (g), dim3 (B1_THREADS), 0, st, (const double *) V, ld, k, h, w, n, partial, -1.0);
   hipLaunchKernelGGL (sum_partials_kernel, dim3 (1), dim3 (NKP_WAVE), 0, st, partial, g, out_nrm2);
}

void launch_axpy_multi (const double *Z, int64_t ld, int k, const double *c, double *x, int64_t n, hipStream_t st)
{
   const int g = red_grid (n);
   hipLaunchKernelGGL ((update_w_kernel<double, double2>), dim3 (g), dim3 (B1_THREADS), 0, st, Z, ld, k, c, x, n, (double *) nullptr, 1.0);
}

// ---------------------------------------------------------------- simple streams
// y = alpha x ; optionally also yf = (float) (alpha x)  (f32 copy of a Krylov basis vector)
__global__ __launch_bounds__ (B1_THREADS)
void scale_to_kernel (const double *__restrict__ x, const double *__restrict__ alpha, double *__restrict__ y, float *__restrict__ yf, int64_t n)
{
   const double a = alpha[0];
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS * 2;
   for (int64_t i = ((int64_t) blockIdx.x * B1_THREADS + threadIdx.x) * 2; i < n; i += stride) {
      if (i + 1 < n) {
         double2 v = *reinterpret_cast<const double2 *> (x + i);
         v.x *= a; v.y *= a;
         *reinterpret_cast<double2 *> (y + i) = v;
         if (yf) *reinterpret_cast<float2 *> (yf + i) = make_float2 ((float) v.x, (float) v.y);
      } else {
         y[i] = a * x[i];
         if (yf) yf[i] = (float) (a * x[i]);
      }
   }
}

void launch_scale_to (const double *x, const double *alpha_dev, double *y, float *yf, int64_t n, hipStream_t st)
{
   hipLaunchKernelGGL (scale_to_kernel, dim3 (red_grid (n)), dim3 (B1_THREADS), 0, st, x, alpha_dev, y, yf, n);
}

__global__ __launch_bounds__ (B1_THREADS)
void axpby_kernel (double a, const double *__restrict__ x, double b, double *__restrict__ y, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; i < n; i += stride)
      y[i] = (b == 0.0 ? 0.0 : b * y[i]) + (x ? a * x[i] : a);
}

void launch_axpby (double a, const double *x, double b, double *y, int64_t n, hipStream_t st)
{
   hipLaunchKernelGGL (axpby_kernel, dim3 (red_grid (n) * 2), dim3 (B1_THREADS), 0, st, a, x, b, y, n);
}

__global__ __launch_bounds__ (B1_THREADS)
void vmul_kernel (const double *x, const double *__restrict__ w, double *y, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; i < n; i += stride) y[i] = x[i] * w[i];
}

void launch_vmul (const double *x, const double *w, double *y, int64_t n, hipStream_t st)
{
   hipLaunchKernelGGL (vmul_kernel, dim3 (red_grid (n) * 2), dim3 (B1_THREADS), 0, st, x, w, y, n);
}

void launch_copy (const double *x, double *y, int64_t n, hipStream_t st)
{
   (void) hipMemcpyAsync (y, x, (size_t) n * sizeof (double), hipMemcpyDeviceToDevice, st);
}

// zero fill in ONE launch: hipMemsetAsync splits a buffer whose size is not a multiple of its block into two kernels (an aligned
// body and a 256-thread remainder), ~10 us of launch floors per level of the cycle instead of ~5
__global__ __launch_bounds__ (B1_THREADS)
void zero_kernel (double *__restrict__ y, int64_t n)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS * 2;
   for (int64_t i = ((int64_t) blockIdx.x * B1_THREADS + threadIdx.x) * 2; i < n; i += stride) {
      if (i + 1 < n) *reinterpret_cast<double2 *> (y + i) = make_double2 (0.0, 0.0);       // vectors are 16-byte aligned (hipMalloc, even offsets)
      else y[i] = 0.0;
   }
}

void launch_fill (double *y, double v, int64_t n, hipStream_t st)
{
   if (n <= 0) return;
   if (v == 0.0) {
      if ((reinterpret_cast<uintptr_t> (y) & 15) == 0) hipLaunchKernelGGL (zero_kernel, dim3 (red_grid (n) * 2), dim3 (B1_THREADS), 0, st, y, n);
      else (void) hipMemsetAsync (y, 0, (size_t) n * sizeof (double), st);
   } else
      launch_axpby (v, nullptr, 0.0, y, n, st);
}

// ---------------------------------------------------------------- dot, berr
__global__ __launch_bounds__ (B1_THREADS)
void dot_kernel (const double *__restrict__ x, const double *__restrict__ y, int64_t n, double *__restrict__ partial)
{
   __shared__ double sh[B1_THREADS / NKP_WAVE];
   double acc = 0.0;
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS * 2;
   for (int64_t i = ((int64_t) blockIdx.x * B1_THREADS + threadIdx.x) * 2; i < n; i += stride) {
      if (i + 1 < n) {
         const double2 a = *reinterpret_cast<const double2 *> (x + i);
         const double2 b = *reinterpret_cast<const double2 *> (y + i);
         acc += a.x * b.x + a.y * b.y;
      } else
         acc += x[i] * y[i];
   }
   const double s = block_sum (acc, sh);
   if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

void launch_dot (const double *x, const double *y, int64_t n, double *partial, double *out, hipStream_t st)
{
   const int g = red_grid (n);
   hipLaunchKernelGGL (dot_kernel, dim3 (g), dim3 (B1_THREADS), 0, st, x, y, n, partial);
   hipLaunchKernelGGL (sum_partials_kernel, dim3 (1), dim3 (NKP_WAVE), 0, st, partial, g, out);
}

__global__ __launch_bounds__ (B1_THREADS)
void berr_kernel (const double *__restrict__ r, const double *__restrict__ den, int64_t n, double *__restrict__ partial)
{
   __shared__ double sh[B1_THREADS / NKP_WAVE];
   double m = 0.0;
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; i < n; i += stride) {
      const double a = fabs (r[i]), d = den[i];
      const double q = (d > 0.0) ? a / d : (a > 0.0 ? 1.0e300 : 0.0);
      m = q > m ? q : m;
   }
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) {
      const double o = __shfl_down (m, off);
      m = o > m ? o : m;
   }
   const int lane = threadIdx.x & (NKP_WAVE - 1), wv = threadIdx.x / NKP_WAVE;
   if (lane == 0) sh[wv] = m;
   __syncthreads ();
   if (threadIdx.x == 0) {
      for (int w = 1; w < B1_THREADS / NKP_WAVE; w++) m = sh[w] > m ? sh[w] : m;
      partial[blockIdx.x] = m;
   }
}

__global__ __launch_bounds__ (NKP_WAVE)
void max_partials_kernel (const double *__restrict__ partial, int nblk, double *__restrict__ out)
{
   double m = 0.0;
   for (int b = threadIdx.x; b < nblk; b += NKP_WAVE) m = partial[b] > m ? partial[b] : m;
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) {
      const double o = __shfl_down (m, off);
      m = o > m ? o : m;
   }
   if (threadIdx.x == 0) out[0] = m;
}

void launch_berr (const double *r, const double *den, int64_t n, double *partial, double *out, hipStream_t st)
{
   const int g = red_grid (n);
   hipLaunchKernelGGL (berr_kernel, dim3 (g), dim3 (B1_THREADS), 0, st, r, den, n, partial);
   hipLaunchKernelGGL (max_partials_kernel, dim3 (1), dim3 (NKP_WAVE), 0, st, partial, g, out);
}

// ---------------------------------------------------------------- Hessenberg column epilogue
__global__ __launch_bounds__ (NKP_WAVE)
void finish_column_kernel (double *__restrict__ h, const double *__restrict__ h2, int k,
                           const double *__restrict__ nrm2, double *__restrict__ inv)
{
   if (h2)
      for (int j = threadIdx.x; j < k; j += NKP_WAVE) h[j] += h2[j];
   if (threadIdx.x == 0) {
      const double t = sqrt (nrm2[0]);
      h[k] = t;
      inv[0] = (t > 0.0) ? 1.0 / t : 0.0;
   }
}

// distributed flavour: ||w - V h||^2 = w.w - sum h_i^2 from the ALREADY reduced multi-dot message (h[0..k-1], h[k] = w.w), so
// that an Arnoldi step needs one allreduce instead of two.  The difference loses log10 (w.w / result) digits; below 1e-8 of
// w.w (the new direction is numerically inside the old space) h[k] is returned NEGATIVE: the host takes its magnitude and ends
// the restart cycle there, and the true residual of the restart decides.  One wave, fixed order.
__global__ __launch_bounds__ (NKP_WAVE)
void finish_column_pythagoras_kernel (double *__restrict__ h, int k, double *__restrict__ inv)
{
   double s = 0.0;
   for (int j = threadIdx.x; j < k; j += NKP_WAVE) s += h[j] * h[j];
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) s += __shfl_down (s, off);
   if (threadIdx.x == 0) {
      const double ww = h[k];
      double t2 = ww - s;
      const bool weak = !(t2 >= 1e-8 * ww);
      if (!(t2 > 0.0)) t2 = 0.0;
      const double t = sqrt (t2);
      h[k] = weak ? -t : t;
      inv[0] = (t > 0.0) ? 1.0 / t : 0.0;
   }
}

void launch_finish_column_pythagoras (double *h, int k, double *inv, hipStream_t st)
{
   hipLaunchKernelGGL (finish_column_pythagoras_kernel, dim3 (1), dim3 (NKP_WAVE), 0, st, h, k, inv);
}

void launch_finish_column (double *h, const double *h2, int k, const double *nrm2, double *inv, hipStream_t st)
{
   hipLaunchKernelGGL (finish_column_kernel, dim3 (1), dim3 (NKP_WAVE), 0, st, h, h2, k, nrm2, inv);
}

// ---------------------------------------------------------------- grid transfer / permutation
__global__ __launch_bounds__ (B1_THREADS)
void restrict_sum_kernel (const int *__restrict__ rptr, const int *__restrict__ ridx, const double *__restrict__ fine,
                          double *__restrict__ coarse, int64_t nc)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t I = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; I < nc; I += stride) {
      double acc = 0.0;
      for (int q = rptr[I]; q < rptr[I + 1]; q++) acc += fine[ridx[q]];
      coarse[I] = acc;
   }
}

void launch_restrict_sum (const int *rptr, const int *ridx, const double *fine, double *coarse, int64_t nc, hipStream_t st)
{
   if (nc <= 0) return;
   hipLaunchKernelGGL (restrict_sum_kernel, dim3 (red_grid (nc) * 2), dim3 (B1_THREADS), 0, st, rptr, ridx, fine, coarse, nc);
}

__global__ __launch_bounds__ (B1_THREADS)
void prolong_add_kernel (const int *__restrict__ cmap, const double *__restrict__ coarse, double *__restrict__ fine, int64_t nf, double omega)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; i < nf; i += stride) fine[i] += omega * coarse[cmap[i]];
}

// fine += omega * P coarse (omega = 1: the plain piecewise-constant prolongation; x * 1.0 is exact)
void launch_prolong_add (const int *cmap, const double *coarse, double *fine, int64_t nf, double omega, hipStream_t st)
{
   if (nf <= 0) return;
   hipLaunchKernelGGL (prolong_add_kernel, dim3 (red_grid (nf) * 2), dim3 (B1_THREADS), 0, st, cmap, coarse, fine, nf, omega);
}

__global__ __launch_bounds__ (B1_THREADS)
void gather_kernel (const int *__restrict__ perm, const double *__restrict__ in, double *__restrict__ out, int64_t n, int scatter)
{
   const int64_t stride = (int64_t) gridDim.x * B1_THREADS;
   for (int64_t i = (int64_t) blockIdx.x * B1_THREADS + threadIdx.x; i < n; i += stride) {
      if (scatter) out[perm[i]] = in[i];
      else out[i] = in[perm[i]];
   }
}

void launch_gather (const int *perm, const double *in, double *out, int64_t n, hipStream_t st)
{
   if (n <= 0) return;
   hipLaunchKernelGGL (gather_kernel, dim3 (red_grid (n) * 2), dim3 (B1_THREADS), 0, st, perm, in, out, n, 0);
}

void launch_scatter (const int *perm, const double *in, double *out, int64_t n, hipStream_t st)
{
   if (n <= 0) return;
   hipLaunchKernelGGL (gather_kernel, dim3 (red_grid (n) * 2), dim3 (B1_THREADS), 0, st, perm, in, out, n, 1);
}

// one wave per output row; the coarsest level is at most a few thousand unknowns
__global__ __launch_bounds__ (B1_THREADS)
void dense_matvec_kernel (const double *__restrict__ M, const double *__restrict__ x, double *__restrict__ y, int n)
{
   const int row = (int) ((blockIdx.x * B1_THREADS + threadIdx.x) / NKP_WAVE);
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   if (row >= n) return;
   const double *m = M + (int64_t) row * n;
   double acc = 0.0;
   for (int c = lane; c < n; c += NKP_WAVE) acc += m[c] * x[c];
   acc = wave_sum (acc);
   if (lane == 0) y[row] = acc;
}

void launch_dense_matvec (const double *Minv, const double *x, double *y, int n, hipStream_t st)
{
   if (n <= 0) return;
   const int waves_per_block = B1_THREADS / NKP_WAVE;
   hipLaunchKernelGGL (dense_matvec_kernel, dim3 ((n + waves_per_block - 1) / waves_per_block), dim3 (B1_THREADS), 0, st, Minv, x, y, n);
}
