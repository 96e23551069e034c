// The small end of the V-cycle in ONE launch.
//
// A 1 degree cycle issues ~150 kernels; on the last levels (a few thousand rows) every one of them -- residual rows,
// column solves, restriction, prolongation -- does microseconds of work behind a 7-12 us launch-to-finish floor, so two
// levels that hold 0.1 % of the rows cost 15 % of the cycle (rocprof: 28 launches per level per cycle).  Here ONE
// workgroup of 1024 threads walks the whole sub-cycle of the levels >= l_tail: pre-smoothing, residual, restriction
// down to the dense coarsest solve, then prolongation and post-smoothing back up, with workgroup barriers where the
// multi-kernel path has kernel boundaries.  A single workgroup needs no grid-wide barrier, so there is nothing that
// can deadlock: every thread reaches every __syncthreads.
//
// Same arithmetic in the same order as the kernels it replaces (thread-per-row residuals add the row's products in
// stored order like the CSR-stream kernels do, thread-per-column substitutions follow colblock_apply_lanes_kernel, the
// dense solve reduces per wave like dense_matvec_kernel), so the cycle's result does not change by a bit
// (tests/test_gpu_parity.py::test_tail_kernel_is_bit_identical).
#include "nkp_dev.h"
#include "multilevel.h"

#define TAIL_THREADS 1024
#define TAIL_MAX_LEVELS 8

struct TailLevel {
   int n, rows0, nc, nu;
   const int *rowptr, *colind;
   const float *valf;
   const double *val;
   int grp[3];                        // first group of colour 0, of colour 1, end
   int P, gw, ngrp;
   const int *grp_nb, *grp_maxlen, *grp_row0, *col_slot;
   const long long *grp_base;
   const float *fac_tf;
   const double *fac_t;
   const int *cmap, *rptr, *ridx;
   double *x, *b, *r;
};

struct TailArgs {
   int nlev;                          // levels in the tail, the last one is the dense coarsest level
   double omega;
   const double *coarse_inv;
   const float *coarse_invf;      // f32 storage mode: the copy the cycle multiplies with (rows padded to coarse_ldf), else NULL
   int coarse_ldf;
   TailLevel lev[TAIL_MAX_LEVELS];
};

namespace {

__device__ __forceinline__ double tail_wave_sum (double v)
{
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) v += __shfl_down (v, off);
   return v;
}

// rows [r0, r1): out = b - L x  (thread per row, products added in stored order)
__device__ void tail_residual (const TailLevel &V, int r0, int r1)
{
   for (int row = r0 + (int) threadIdx.x; row < r1; row += TAIL_THREADS) {
      double acc = 0.0;
      const int e1 = V.rowptr[row + 1];
      if (V.valf)
         for (int e = V.rowptr[row]; e < e1; e++) acc += (double) V.valf[e] * V.x[V.colind[e]];
      else
         for (int e = V.rowptr[row]; e < e1; e++) acc += V.val[e] * V.x[V.colind[e]];
      V.r[row] = V.b[row] - acc;
   }
   __syncthreads ();
}

// column solves of colour c: x_c = (accumulate ? x_c : 0) + B_c^-1 rhs_c; rhs is V.r or V.b; thread per column
template <int P, class FT>
__device__ void tail_columns_t (const TailLevel &V, int c, const double *rhs, const FT *fac, int accumulate)
{
   const int g0 = V.grp[c], ncols = (V.grp[c + 1] - g0) * V.gw;
   for (int t = threadIdx.x; t < ncols; t += TAIL_THREADS) {
      const int g = g0 + t / V.gw, lane = t % V.gw;
      if (lane >= V.grp_nb[g]) continue;
      const int ml = V.grp_maxlen[g];
      const int row0 = V.grp_row0[g] + V.col_slot[g * V.gw + lane];
      const int len = V.col_slot[(V.ngrp + g) * V.gw + lane];
      const FT *ft = fac + V.grp_base[g] + lane;
      const int dstride = ml * V.gw;
      double *z = V.r;                               // scratch: the colour's rows of r are dead once read
      double carry[P];
#pragma unroll
      for (int q = 0; q < P; q++) carry[q] = 0.0;
      for (int k0 = 0; k0 < ml; k0 += 8) {
         double v[8];
#pragma unroll
         for (int j = 0; j < 8; j++) v[j] = (k0 + j < len) ? rhs[row0 + k0 + j] : 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++) {
            const int k = k0 + j;
            double y = v[j];
#pragma unroll
            for (int q = P; q >= 1; q--) {
               const double prev = (j - q >= 0) ? v[j - q >= 0 ? j - q : 0] : carry[q - j - 1 >= 0 && q - j - 1 < P ? q - j - 1 : 0];
               if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * V.gw] * prev;
            }
            v[j] = y;
         }
#pragma unroll
         for (int j = 0; j < 8; j++)
            if (k0 + j < len) z[row0 + k0 + j] = v[j];
#pragma unroll
         for (int q = 1; q <= P; q++) carry[q - 1] = v[8 - q];
      }
      double nxt[P];
#pragma unroll
      for (int q = 0; q < P; q++) nxt[q] = 0.0;
      for (int k0 = ml - 8; k0 >= 0; k0 -= 8) {
         double v[8];
#pragma unroll
         for (int j = 0; j < 8; j++) v[j] = (k0 + j < len) ? z[row0 + k0 + j] : 0.0;
#pragma unroll
         for (int j = 7; j >= 0; j--) {
            const int k = k0 + j;
            double x = v[j];
#pragma unroll
            for (int q = P; q >= 1; q--) {
               const double nv = (j + q <= 7) ? v[j + q <= 7 ? j + q : 7] : nxt[j + q - 8 >= 0 && j + q - 8 < P ? j + q - 8 : 0];
               if (k + q < ml) x -= (double) ft[(P + q) * dstride + k * V.gw] * nv;
            }
            x *= (double) ft[P * dstride + k * V.gw];
            v[j] = x;
         }
#pragma unroll
         for (int j = 0; j < 8; j++)
            if (k0 + j < len) {
               const int row = row0 + k0 + j;
               V.x[row] = accumulate ? V.x[row] + v[j] : v[j];
            }
#pragma unroll
         for (int q = 0; q < P; q++) nxt[q] = v[q];
      }
   }
   __syncthreads ();
}

__device__ void tail_columns (const TailLevel &V, int c, const double *rhs, int accumulate)
{
   if (V.fac_tf) {
      if (V.P == 1) tail_columns_t<1, float> (V, c, rhs, V.fac_tf, accumulate);
      else if (V.P == 2) tail_columns_t<2, float> (V, c, rhs, V.fac_tf, accumulate);
      else tail_columns_t<4, float> (V, c, rhs, V.fac_tf, accumulate);
   } else {
      if (V.P == 1) tail_columns_t<1, double> (V, c, rhs, V.fac_t, accumulate);
      else if (V.P == 2) tail_columns_t<2, double> (V, c, rhs, V.fac_t, accumulate);
      else tail_columns_t<4, double> (V, c, rhs, V.fac_t, accumulate);
   }
}

__device__ void tail_half_sweep (const TailLevel &V, int c)
{
   const int r0 = c == 0 ? 0 : V.rows0, r1 = c == 0 ? V.rows0 : V.n;
   tail_residual (V, r0, r1);
   tail_columns (V, c, V.r, 1);
}

__global__ __launch_bounds__ (TAIL_THREADS)
void ml_tail_kernel (TailArgs A)
{
   const int last = A.nlev - 1;
   for (int l = 0; l < last; l++) {
      const TailLevel &V = A.lev[l];
      for (int i = threadIdx.x; i < V.n; i += TAIL_THREADS) V.x[i] = 0.0;
      __syncthreads ();
      tail_columns (V, 0, V.b, 0);                 // first half sweep from x = 0: r = b on colour 0
      tail_half_sweep (V, 1);
      for (int s = 1; s < V.nu; s++) { tail_half_sweep (V, 0); tail_half_sweep (V, 1); }
      tail_residual (V, 0, V.n);
      const TailLevel &C = A.lev[l + 1];
      for (int I = threadIdx.x; I < V.nc; I += TAIL_THREADS) {
         double acc = 0.0;
         for (int q = V.rptr[I]; q < V.rptr[I + 1]; q++) acc += V.r[V.ridx[q]];
         C.b[I] = acc;
      }
      __syncthreads ();
   }
   {
      // dense coarsest solve: one wave per row, like dense_matvec_kernel
      const TailLevel &V = A.lev[last];
      const int lane = threadIdx.x & (NKP_WAVE - 1), wave = threadIdx.x / NKP_WAVE;
      for (int row = wave; row < V.n; row += TAIL_THREADS / NKP_WAVE) {
         double acc = 0.0;
         if (A.coarse_invf) {
            // f32 copy of the inverse, summed like dense_matvec_f32_kernel (dense.hip): four consecutive columns per lane and step
            const float4 *m4 = reinterpret_cast<const float4 *> (A.coarse_invf + (int64_t) row * A.coarse_ldf);
            for (int c4 = lane; c4 * 4 < V.n; c4 += NKP_WAVE) {
               const float4 v = m4[c4];
               const int c = c4 * 4;
               acc += (double) v.x * V.b[c];
               if (c + 1 < V.n) acc += (double) v.y * V.b[c + 1];
               if (c + 2 < V.n) acc += (double) v.z * V.b[c + 2];
               if (c + 3 < V.n) acc += (double) v.w * V.b[c + 3];
            }
         } else {
            const double *m = A.coarse_inv + (int64_t) row * V.n;
            for (int c = lane; c < V.n; c += NKP_WAVE) acc += m[c] * V.b[c];
         }
         acc = tail_wave_sum (acc);
         if (lane == 0) V.x[row] = acc;
      }
      __syncthreads ();
   }
   for (int l = last - 1; l >= 0; l--) {
      const TailLevel &V = A.lev[l];
      const TailLevel &C = A.lev[l + 1];
      for (int i = threadIdx.x; i < V.n; i += TAIL_THREADS) V.x[i] += A.omega * C.x[V.cmap[i]];
      __syncthreads ();
      for (int s = 0; s < V.nu; s++) { tail_half_sweep (V, 1); tail_half_sweep (V, 0); }
   }
}

}  // namespace

// levels [l0, end) of H in one launch; the caller has put the restricted residual into lev[l0].b and reads lev[l0].x
int ml_tail_launch (MlHierarchy &H, int l0, hipStream_t st)
{
   const int nlev = (int) H.lev.size () - l0;
   if (nlev < 1 || nlev > TAIL_MAX_LEVELS) return 1;
   TailArgs A;
   A.nlev = nlev;
   A.omega = H.omega;
   A.coarse_inv = H.coarse_inv;
   A.coarse_invf = H.coarse_invf;
   A.coarse_ldf = H.coarse_ldf;
   for (int q = 0; q < nlev; q++) {
      MlLevel &V = H.lev[l0 + q];
      TailLevel &T = A.lev[q];
      T.n = (int) V.n; T.rows0 = (int) V.rows0; T.nc = (int) V.nc;
      T.nu = (l0 + q >= H.coarse_from) ? H.nu_coarse : H.nu;
      T.rowptr = V.L.rowptr; T.colind = V.L.colind; T.valf = V.L.valf; T.val = V.L.val;
      T.grp[0] = V.color_grp[0]; T.grp[1] = V.color_grp[1]; T.grp[2] = V.color_grp[2];
      T.P = V.B.P; T.gw = V.B.gw; T.ngrp = V.B.ngrp;
      T.grp_nb = V.B.grp_nb; T.grp_maxlen = V.B.grp_maxlen; T.grp_row0 = V.B.grp_row0; T.col_slot = V.B.col_slot;
      T.grp_base = V.B.grp_base; T.fac_tf = V.B.fac_tf; T.fac_t = V.B.fac_t;
      T.cmap = V.cmap; T.rptr = V.rptr; T.ridx = V.ridx;
      T.x = V.x; T.b = V.b; T.r = V.r;
      V.cur[0] = V.cur[1] = 0;
   }
   hipLaunchKernelGGL (ml_tail_kernel, dim3 (1), dim3 (TAIL_THREADS), 0, st, A);
   return 0;
}
