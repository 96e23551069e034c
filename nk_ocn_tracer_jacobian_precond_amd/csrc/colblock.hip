// Water-column block preconditioner for gfx950: ONE WATER COLUMN PER WAVEFRONT.
//
// The flat state vector enumerates j outer, i middle, k inner (reference src/matrix.c:239-251),
// so the levels k = 0..KMT-1 of one column are contiguous rows and the within-column coupling
// (k+-1 from vertical mixing/advection src/matrix.c:808-819, k+-2 with upwind3 :843-854) is a
// narrow band.  Lane l of a wave owns level l of the column (two levels per lane when a block
// is longer than 64, e.g. km = 80), the band lives in registers, and the sequential
// elimination/substitution over k broadcasts one value per step with v_readlane (the step
// index is wave-uniform, so no LDS round trip and no bpermute).
//
// Exact LU of the band == ILU(0) of the column block (no fill leaves the band), so this one
// kernel pair is both the "block-Jacobi" and the "column-ILU(0)" preconditioner; it stands in
// for SuperLU's pdgstrf + pdgstrs_Bglobal (reference src/SuperLU_brief_tree.txt:8-17).
//
// Roofline: HBM-bound.  Apply traffic = (2P+1)*8*n factor bytes + 16*n vector bytes.
#include "nkp_dev.h"

#define CB_THREADS 256
#define CB_WAVES (CB_THREADS / NKP_WAVE)

__device__ __forceinline__ double readlane_f64 (double v, int lane)
{
   int lo = __double2loint (v), hi = __double2hiint (v);
   lo = __builtin_amdgcn_readlane (lo, lane);
   hi = __builtin_amdgcn_readlane (hi, lane);
   return __hiloint2double (hi, lo);
}

__device__ __forceinline__ int wave_block_id ()
{
   // wave-uniform by construction; readfirstlane tells the compiler so (scalar loads, SGPR loop bounds)
   return __builtin_amdgcn_readfirstlane ((int) ((blockIdx.x * CB_THREADS + threadIdx.x) / NKP_WAVE));
}

// ---------------------------------------------------------------- measure
// out[0] = max in-block |col - row| ; out[1] = rows whose diagonal entry is missing or zero ;
// out[2] = longest block
__global__ __launch_bounds__ (CB_THREADS)
void colblock_measure_kernel (const int *__restrict__ rowptr, const int *__restrict__ colind,
                              const double *__restrict__ val, const int *__restrict__ blk_start, int nblk, int *out)
{
   const int b = wave_block_id ();
   if (b >= nblk) return;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   const int r0 = blk_start[b], r1 = blk_start[b + 1];
   int bw = 0, nodiag = 0;
   for (int r = r0 + lane; r < r1; r += NKP_WAVE) {
      bool have = false;
      for (int e = rowptr[r]; e < rowptr[r + 1]; e++) {
         const int c = colind[e];
         if (c >= r0 && c < r1) {
            const int d = c > r ? c - r : r - c;
            bw = d > bw ? d : bw;
            if (c == r && val[e] != 0.0) have = true;
         }
      }
      nodiag += have ? 0 : 1;
   }
   for (int off = NKP_WAVE / 2; off > 0; off >>= 1) {
      const int o = __shfl_down (bw, off);
      bw = o > bw ? o : bw;
      nodiag += __shfl_down (nodiag, off);
   }
   if (lane == 0) {
      atomicMax (&out[0], bw);
      if (nodiag) atomicAdd (&out[1], nodiag);
      atomicMax (&out[2], r1 - r0);
   }
}

// ---------------------------------------------------------------- extract + factor
template <int P, int RPL>
__global__ __launch_bounds__ (CB_THREADS)
void colblock_factor_kernel (const int *__restrict__ rowptr, const int *__restrict__ colind,
                             const double *__restrict__ val, const int *__restrict__ blk_start, int nblk,
                             int64_t n, double *__restrict__ fac, int *status, int *dropped)
{
   const int b = wave_block_id ();
   if (b >= nblk) return;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   const int r0 = blk_start[b];
   const int len = blk_start[b + 1] - r0;

   double a[RPL][2 * P + 1];
   int drop = 0;
#pragma unroll
   for (int s = 0; s < RPL; s++) {
#pragma unroll
      for (int d = 0; d <= 2 * P; d++) a[s][d] = 0.0;
      const int li = s * NKP_WAVE + lane;
      if (li < len) {
         const int r = r0 + li;
         for (int e = rowptr[r]; e < rowptr[r + 1]; e++) {
            const int c = colind[e];
            if (c < r0 || c >= r0 + len) continue;
            const int d = c - r;
            const double v = val[e];
            if (d < -P || d > P) { drop = 1; continue; }
#pragma unroll
            for (int dd = -P; dd <= P; dd++)
               if (d == dd) a[s][dd + P] = v;
         }
      }
   }

   // right-looking banded LU without pivoting; step k is wave-uniform
   int bad = 0;
   for (int k = 0; k < len; k++) {
      const int ks = k >> 6, kl = k & (NKP_WAVE - 1);
      double u[P + 1];
#pragma unroll
      for (int s = 0; s < RPL; s++)
         if (ks == s) {
#pragma unroll
            for (int q = 0; q <= P; q++) u[q] = readlane_f64 (a[s][P + q], kl);
         }
      if (!(fabs (u[0]) > 1.0e-300) && bad == 0) bad = r0 + k + 1;
      const double inv = 1.0 / u[0];
#pragma unroll
      for (int dist = 1; dist <= P; dist++) {
         const int kr = k + dist;
         if (kr < len) {
            const int ts = kr >> 6, tl = kr & (NKP_WAVE - 1);
#pragma unroll
            for (int s = 0; s < RPL; s++)
               if (ts == s && lane == tl) {
                  const double l = a[s][P - dist] * inv;
                  a[s][P - dist] = l;
#pragma unroll
                  for (int q = 1; q <= P; q++)
                     if (P - dist + q <= 2 * P) a[s][P - dist + q] -= l * u[q];
               }
         }
      }
   }

#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      if (li < len) {
         const int64_t r = r0 + li;
         a[s][P] = 1.0 / a[s][P];
#pragma unroll
         for (int d = 0; d <= 2 * P; d++) fac[(int64_t) d * n + r] = a[s][d];
      }
   }
   if (lane == 0 && bad) atomicCAS (status, 0, bad);
   if (drop) *dropped = 1;
}

// ---------------------------------------------------------------- apply  z = (LU)^-1 r
template <int P, int RPL>
__global__ __launch_bounds__ (CB_THREADS)
void colblock_apply_kernel (const int *__restrict__ blk_start, int b_first, int nblk, int64_t n,
                            const double *__restrict__ fac, const double *__restrict__ rhs, double *__restrict__ z, int accumulate)
{
   const int b = wave_block_id () + b_first;
   if (b >= nblk) return;
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   const int r0 = blk_start[b];
   const int len = blk_start[b + 1] - r0;

   double y[RPL], invd[RPL], L[RPL][P], U[RPL][P];
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      y[s] = 0.0;
      invd[s] = 0.0;
#pragma unroll
      for (int q = 0; q < P; q++) { L[s][q] = 0.0; U[s][q] = 0.0; }
      if (li < len) {
         const int64_t r = r0 + li;
         y[s] = rhs[r];
         invd[s] = fac[(int64_t) P * n + r];
#pragma unroll
         for (int q = 1; q <= P; q++) {
            L[s][q - 1] = fac[(int64_t) (P - q) * n + r];     // l(r, r-q)
            U[s][q - 1] = fac[(int64_t) (P + q) * n + r];     // u(r, r+q)
         }
      }
   }

   // forward: y <- L^-1 y   (unit lower band)
   for (int k = 0; k < len - 1; k++) {
      const int ks = k >> 6, kl = k & (NKP_WAVE - 1);
      double yk = 0.0;
#pragma unroll
      for (int s = 0; s < RPL; s++)
         if (ks == s) yk = readlane_f64 (y[s], kl);
#pragma unroll
      for (int s = 0; s < RPL; s++) {
         const int rel = s * NKP_WAVE + lane - k;               // my level minus k
#pragma unroll
         for (int q = 1; q <= P; q++)
            if (rel == q) y[s] -= L[s][q - 1] * yk;
      }
   }
   // backward: y <- U^-1 y
   for (int k = len - 1; k >= 0; k--) {
      const int ks = k >> 6, kl = k & (NKP_WAVE - 1);
      double xk = 0.0;
#pragma unroll
      for (int s = 0; s < RPL; s++)
         if (ks == s) {
            if (lane == kl) y[s] *= invd[s];
            xk = readlane_f64 (y[s], kl);
         }
#pragma unroll
      for (int s = 0; s < RPL; s++) {
         const int rel = k - (s * NKP_WAVE + lane);             // k minus my level
#pragma unroll
         for (int q = 1; q <= P; q++)
            if (rel == q) y[s] -= U[s][q - 1] * xk;
      }
   }
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      if (li < len) {
         if (accumulate) z[(int64_t) r0 + li] += y[s];
         else z[(int64_t) r0 + li] = y[s];
      }
   }
}

// ---------------------------------------------------------------- launchers
static inline dim3 cb_grid (int nblk) { return dim3 ((nblk + CB_WAVES - 1) / CB_WAVES); }

void launch_colblock_measure (const CsrDev &A, const ColBlocksDev &B, int *d_out3, hipStream_t st)
{
   if (B.nblk == 0) return;
   hipLaunchKernelGGL (colblock_measure_kernel, cb_grid (B.nblk), dim3 (CB_THREADS), 0, st,
                       A.rowptr, A.colind, A.val, B.blk_start, B.nblk, d_out3);
}

#define CB_DISPATCH(KERNEL, ...) CB_DISPATCH_N (KERNEL, B.nblk, __VA_ARGS__)
#define CB_DISPATCH_N(KERNEL, NB, ...)                                                             \
   do {                                                                                           \
      const int rpl = B.max_len <= NKP_WAVE ? 1 : 2;                                              \
      if (B.P == 1 && rpl == 1) hipLaunchKernelGGL ((KERNEL<1, 1>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__); \
      else if (B.P == 1) hipLaunchKernelGGL ((KERNEL<1, 2>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__);        \
      else if (B.P == 2 && rpl == 1) hipLaunchKernelGGL ((KERNEL<2, 1>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__); \
      else if (B.P == 2) hipLaunchKernelGGL ((KERNEL<2, 2>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__);        \
      else if (rpl == 1) hipLaunchKernelGGL ((KERNEL<4, 1>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__);        \
      else hipLaunchKernelGGL ((KERNEL<4, 2>), cb_grid (NB), dim3 (CB_THREADS), 0, st, __VA_ARGS__);                      \
   } while (0)

void launch_colblock_factor (const CsrDev &A, ColBlocksDev &B, int *d_status, hipStream_t st)
{
   if (B.nblk == 0) return;
   int *d_dropped = d_status + 1;
   CB_DISPATCH (colblock_factor_kernel, A.rowptr, A.colind, A.val, B.blk_start, B.nblk, B.n, B.fac, d_status, d_dropped);
}

void launch_colblock_apply (const ColBlocksDev &B, const double *r, double *z, hipStream_t st)
{
   if (B.nblk == 0) return;
   CB_DISPATCH (colblock_apply_kernel, B.blk_start, 0, B.nblk, B.n, B.fac, r, z, 0);
}

void launch_colblock_apply_range (const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, hipStream_t st)
{
   if (b1 <= b0) return;
   CB_DISPATCH_N (colblock_apply_kernel, b1 - b0, B.blk_start, b0, b1, B.n, B.fac, r, z, accumulate);
}
