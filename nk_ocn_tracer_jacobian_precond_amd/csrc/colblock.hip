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

#include <algorithm>

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
// R32: the factors are rounded to f32 on the way in, which gives exactly the values the f32 lane layouts store, so the
// result is the one of the lane-per-column kernels in the f32-storage mode of the multilevel cycle
template <int P, int RPL, bool R32 = false>
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
         if (R32) invd[s] = (double) (float) invd[s];
#pragma unroll
         for (int q = 1; q <= P; q++) {
            L[s][q - 1] = fac[(int64_t) (P - q) * n + r];     // l(r, r-q)
            U[s][q - 1] = fac[(int64_t) (P + q) * n + r];     // u(r, r+q)
            if (R32) { L[s][q - 1] = (double) (float) L[s][q - 1]; U[s][q - 1] = (double) (float) U[s][q - 1]; }
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

// ---------------------------------------------------------------- fused half sweep, one water column per wave
// Small levels of the multilevel cycle (a few thousand columns) run at the latency floor of their launches: a residual SpMV
// over the colour's rows (~7 us) followed by the wave-per-column solve above (~11 us), both far below one wave per SIMD.
// Here ONE launch does both for one colour: lane l of the wave that owns a column computes the residual of ITS row straight
// from the CSR arrays (row-per-lane: a level of this size sits in L2 / MALL, so the uncoalesced row reads cost latency, not
// bandwidth, and the loads of 8 entries are in flight together), then the wave runs the band substitution on the
// residual it holds in registers, and x_new = x_old + z goes out.  r never touches memory and half of the launches go.
// x comes from TWO buffers like gs_fused_kernel's (rows < split from xa, the others from xb; new values to xout): columns
// of one colour are coupled, so the sweep must not see its own updates (multilevel.hip ping-pongs the buffers).
// Same products, same per-row summation order, same substitution as the two-kernel path => identical bits.
#define GSW_UNROLL 24
#define GSW_CAP 1536          // entries of one water column staged per wave (LDS: 4 waves x 1536 x 8 bytes = 48 KB with f32 values)
#define GSW_STAGE 8
template <int P, int RPL, class VT, bool R32>
__global__ __launch_bounds__ (CB_THREADS)
void gs_wave_kernel (const int *__restrict__ rowptr, const int *__restrict__ colind, const VT *__restrict__ val,
                     const int *__restrict__ blk_start, int b_first, int b_end, int64_t n, const double *__restrict__ fac,
                     const double *__restrict__ xa, const double *__restrict__ xb, int split, const double *__restrict__ b, double *__restrict__ xout,
                     const int4 *__restrict__ desc)
{
   // the (column, value) entries of the wave's water column -- one contiguous stretch of the CSR arrays -- are staged through LDS with
   // coalesced loads; the row-per-lane reads they replace asked the L1 for every 128-byte line about 24 times (lanes 54 bytes
   // apart, one instruction per entry of the row) and bounded the launch: 26 us per half sweep of the 1 degree level 2 (3416 waves)
   extern __shared__ unsigned char gsw_lds[];
   const int wv = threadIdx.x / NKP_WAVE;
   int *sc = reinterpret_cast<int *> (gsw_lds) + wv * GSW_CAP;
   VT *sv = reinterpret_cast<VT *> (gsw_lds + (size_t) CB_WAVES * GSW_CAP * sizeof (int)) + wv * GSW_CAP;
   const int blk = wave_block_id () + b_first;
   const bool act = blk < b_end;                    // (no early return: the whole workgroup meets at the barrier below)
   const int lane = threadIdx.x & (NKP_WAVE - 1);
   // {first row, rows, first entry, entries} of the column in one 16-byte load (desc: built at setup), else through blk_start and rowptr
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

   double y[RPL], xold[RPL], invd[RPL], L[RPL][P], U[RPL][P];
   int e0[RPL], rl[RPL];
   const int e_begin = d4.z, e_total = d4.w;
   const bool staged = e_total <= GSW_CAP;
   if (staged) {
      for (int k0 = 0; k0 < e_total; k0 += NKP_WAVE * GSW_STAGE) {
         int tc[GSW_STAGE];
         VT tv[GSW_STAGE];
#pragma unroll
         for (int u = 0; u < GSW_STAGE; u++) {
            const int k = k0 + u * NKP_WAVE + lane;
            tc[u] = k < e_total ? colind[e_begin + k] : 0;
            tv[u] = k < e_total ? val[e_begin + k] : (VT) 0;
         }
#pragma unroll
         for (int u = 0; u < GSW_STAGE; u++) {
            const int k = k0 + u * NKP_WAVE + lane;
            if (k < e_total) { sc[k] = tc[u]; sv[k] = tv[u]; }
         }
      }
   }
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      y[s] = 0.0; xold[s] = 0.0; invd[s] = 0.0; e0[s] = 0; rl[s] = 0;
#pragma unroll
      for (int q = 0; q < P; q++) { L[s][q] = 0.0; U[s][q] = 0.0; }
      if (li < len) {
         const int64_t r = r0 + li;
         e0[s] = rowptr[r];
         rl[s] = rowptr[r + 1] - e0[s];
         y[s] = b[r];
         xold[s] = (r < split) ? xa[r] : xb[r];
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
   __syncthreads ();
   // residual of this lane's row(s): entries in stored order, GSW_UNROLL of them requested together (from LDS, or one round trip for the (column, value) pairs of a column too long to stage; then one for the gathered x)
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      double acc = 0.0;
      for (int k0 = 0; __any (k0 < rl[s]); k0 += GSW_UNROLL) {
         int cc[GSW_UNROLL];
         VT vv[GSW_UNROLL];
         if (staged) {
            const int off = e0[s] - e_begin + k0;
#pragma unroll
            for (int u = 0; u < GSW_UNROLL; u++) {
               const bool ok = k0 + u < rl[s];
               cc[u] = ok ? sc[off + u] : 0;
               vv[u] = ok ? sv[off + u] : (VT) 0;
            }
         } else {
#pragma unroll
            for (int u = 0; u < GSW_UNROLL; u++) {
               const bool ok = k0 + u < rl[s];
               cc[u] = ok ? colind[e0[s] + k0 + u] : 0;
               vv[u] = ok ? val[e0[s] + k0 + u] : (VT) 0;
            }
         }
         double xv[GSW_UNROLL];
#pragma unroll
         for (int u = 0; u < GSW_UNROLL; u++) xv[u] = (cc[u] < split) ? xa[cc[u]] : xb[cc[u]];
#pragma unroll
         for (int u = 0; u < GSW_UNROLL; u++)
            if (k0 + u < rl[s]) acc += (double) vv[u] * xv[u];
      }
      if (s * NKP_WAVE + lane < len) y[s] -= acc;
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
         const int rel = s * NKP_WAVE + lane - k;
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
         const int rel = k - (s * NKP_WAVE + lane);
#pragma unroll
         for (int q = 1; q <= P; q++)
            if (rel == q) y[s] -= U[s][q - 1] * xk;
      }
   }
#pragma unroll
   for (int s = 0; s < RPL; s++) {
      const int li = s * NKP_WAVE + lane;
      if (li < len) xout[(int64_t) r0 + li] = xold[s] + y[s];
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

// same with the factors rounded to f32 on load (multilevel cycle with f32 storage)
void launch_colblock_apply_range_r32 (const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, hipStream_t st)
{
   if (b1 <= b0) return;
   const int rpl = B.max_len <= NKP_WAVE ? 1 : 2;
   const dim3 grid = cb_grid (b1 - b0);
#define R32_GO(PP, RR) hipLaunchKernelGGL ((colblock_apply_kernel<PP, RR, true>), grid, dim3 (CB_THREADS), 0, st, B.blk_start, b0, b1, B.n, B.fac, r, z, accumulate)
   if (B.P == 1) { if (rpl == 1) R32_GO (1, 1); else R32_GO (1, 2); }
   else if (B.P == 2) { if (rpl == 1) R32_GO (2, 1); else R32_GO (2, 2); }
   else { if (rpl == 1) R32_GO (4, 1); else R32_GO (4, 2); }
#undef R32_GO
}

__global__ __launch_bounds__ (CB_THREADS)
void wave_desc_kernel (const int *__restrict__ rowptr, const int *__restrict__ blk_start, int nblk, int4 *__restrict__ desc)
{
   const int b = blockIdx.x * CB_THREADS + threadIdx.x;
   if (b >= nblk) return;
   const int r0 = blk_start[b], len = blk_start[b + 1] - r0;
   const int e0 = len > 0 ? rowptr[r0] : 0;
   desc[b] = make_int4 (r0, len, e0, len > 0 ? rowptr[r0 + len] - e0 : 0);
}

void launch_build_wave_desc (const CsrDev &L, ColBlocksDev &B, hipStream_t st)
{
   if (B.nblk > 0 && B.wave_desc)
      hipLaunchKernelGGL (wave_desc_kernel, dim3 ((B.nblk + CB_THREADS - 1) / CB_THREADS), dim3 (CB_THREADS), 0, st, L.rowptr, B.blk_start, B.nblk, reinterpret_cast<int4 *> (B.wave_desc));
}

// blocks [b0, b1) of one colour: xout_rows = x_rows + M^-1 (b - L x)_rows in one launch (gs_wave_kernel); r32: factors rounded
// to f32 on load (the f32 storage mode of the cycle).  The level operator is read through L.valf when it has an f32 copy.
void launch_gs_wave (const CsrDev &L, const ColBlocksDev &B, int b0, int b1, const double *xa, const double *xb, int split, const double *b, double *xout,
                     int r32, hipStream_t st)
{
   if (b1 <= b0) return;
   const int rpl = B.max_len <= NKP_WAVE ? 1 : 2;
   const dim3 grid = cb_grid (b1 - b0);
#define GSW_GO(PP, RR, VT_, R32_, VAL_) do { const size_t lds_ = (size_t) CB_WAVES * GSW_CAP * (sizeof (int) + sizeof (VT_));                              \
                                             static bool opted_ = false;                                                                                        \
                                             if (lds_ > 48 * 1024 && !opted_) { (void) hipFuncSetAttribute ((const void *) gs_wave_kernel<PP, RR, VT_, R32_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_); opted_ = true; } \
                                             hipLaunchKernelGGL ((gs_wave_kernel<PP, RR, VT_, R32_>), grid, dim3 (CB_THREADS), lds_, st, L.rowptr, L.colind, VAL_, \
                                                                 B.blk_start, b0, b1, B.n, B.fac, xa, xb, split, b, xout, reinterpret_cast<const int4 *> (B.wave_desc)); } while (0)
#define GSW_PR(PP, RR) do { if (L.valf) { if (r32) GSW_GO (PP, RR, float, true, L.valf); else GSW_GO (PP, RR, float, false, L.valf); } \
                            else { if (r32) GSW_GO (PP, RR, double, true, L.val); else GSW_GO (PP, RR, double, false, L.val); } } while (0)
   if (B.P == 1) { if (rpl == 1) GSW_PR (1, 1); else GSW_PR (1, 2); }
   else if (B.P == 2) { if (rpl == 1) GSW_PR (2, 1); else GSW_PR (2, 2); }
   else { if (rpl == 1) GSW_PR (4, 1); else GSW_PR (4, 2); }
#undef GSW_PR
#undef GSW_GO
}

// ================================================================ lane-per-column apply
// The wave-per-column substitution above is VALU-issue bound: 2*len dependent steps of
// readlane + predicated FMA per column (rocprof: 137 us per half sweep at 1 degree against ~25 us
// of HBM time).  Here one LANE owns one column: a wave stages the right-hand side of 64
// consecutive columns (a contiguous row range) into LDS with coalesced loads, every lane then
// runs its own short recurrence with the factors stored [diagonal][step k][lane] so that each
// step is one coalesced 512-byte load per diagonal, and the result goes back coalesced.
// 64x fewer wave-instructions per column; same operation order as the sequential kernel and
// the CPU restatement used by the tests, so results are bit-identical.
#include <stdlib.h>
#include <vector>

#define LDS_PAD(i) ((i) + ((i) >> 5))       // break the lane stride (column length) bank pattern

__global__ __launch_bounds__ (NKP_WAVE)
void colblock_transpose_kernel (const int *__restrict__ blk_start, const int *__restrict__ grp_b0, const int *__restrict__ grp_nb,
                                const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int ndiag, int64_t n,
                                const double *__restrict__ fac, double *__restrict__ fac_t, int gw, float *__restrict__ fac_tf, int pack,
                                const int *__restrict__ grp_cols /* NULL: the group is the nb consecutive columns from b0 */)
{
   const int g = blockIdx.x;
   const int lane = threadIdx.x;
   const int b0 = grp_b0[g], nb = grp_nb[g], ml = grp_maxlen[g];
   const long long base = grp_base[g];
   int r0 = 0, len = 0;
   if (lane < nb) {
      const int col = grp_cols ? grp_cols[g * gw + lane] : b0 + lane;
      r0 = blk_start[col];
      len = blk_start[col + 1] - r0;
   }
   if (lane >= gw) return;
   for (int d = 0; d < ndiag; d++)
      for (int k = 0; k < ml; k++)
      {
         const double v = (k < len) ? fac[(int64_t) d * n + r0 + k] : 0.0;
         // pack > 1 (colblock_apply_ldspack_kernel): `pack` consecutive steps of one column side by side, one 16-byte load
         const int64_t at = pack > 1 ? base + (((int64_t) d * (ml / pack) + k / pack) * gw + lane) * pack + (k % pack) : base + ((int64_t) d * ml + k) * gw + lane;
         if (fac_tf) fac_tf[at] = (float) v;
         else fac_t[at] = v;
      }
}

template <int P, int MAXL, class FT>
__global__ __launch_bounds__ (NKP_WAVE)
void colblock_apply_lanes_kernel (const int *__restrict__ blk_start, const int *__restrict__ grp_b0, const int *__restrict__ grp_nb,
                                  const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                  const FT *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                  int gw, int rhs_slots, const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];            // [rhs_slots] staged right-hand side | [(2P+1)*ml*gw] the group's factors
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   // every index this wave needs depends on g alone: one round trip, not a chain through blk_start
   const int nb = grp_nb[g], ml = grp_maxlen[g];
   const int R0 = grp_row0[g], nrows = grp_row0[ngrp + g];
   int s_pre = 0, len_pre = 0;
   if (lane < gw) { s_pre = col_slot[g * gw + lane]; len_pre = col_slot[(ngrp + g) * gw + lane]; }
   FT *fl = reinterpret_cast<FT *> (lds + rhs_slots);
   // the accumulate target is requested together with the right-hand side
   double tz[8];
#pragma unroll
   for (int u = 0; u < 8; u++) {
      const int i = lane + u * NKP_WAVE;
      tz[u] = (accumulate && i < nrows) ? z[(int64_t) R0 + i] : 0.0;
   }
   // bulk, fully coalesced staging: every load is independent, so the whole group is in flight at once
   {
      // 16-byte units of the group's factor block (ml*gw is a multiple of 64, so this is exact for f32 too)
      const double2 *src = reinterpret_cast<const double2 *> (fac_t + grp_base[g]);
      double2 *dst = reinterpret_cast<double2 *> (fl);
      const int cnt2 = (int) (((size_t) (2 * P + 1) * ml * gw * sizeof (FT)) >> 4);
      // the right-hand side and a batch of 20 factor loads per lane are all in flight before the first LDS
      // store waits on them (a group of 8 columns x 64 levels x 5 diagonals is exactly one batch)
      double tr[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         tr[u] = (i < nrows) ? rhs[(int64_t) R0 + i] : 0.0;
      }
      constexpr int BATCH = sizeof (FT) == 4 ? 10 : 20;      // 16-byte loads per lane: one batch covers 8 x 64 x 5 factors
      for (int i0 = lane; i0 < cnt2; i0 += BATCH * NKP_WAVE) {
         double2 t[BATCH];
#pragma unroll
         for (int u = 0; u < BATCH; u++) {
            const int i = i0 + u * NKP_WAVE;
            t[u] = (i < cnt2) ? src[i] : make_double2 (0.0, 0.0);
         }
#pragma unroll
         for (int u = 0; u < BATCH; u++) {
            const int i = i0 + u * NKP_WAVE;
            if (i < cnt2) dst[i] = t[u];
         }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         if (i < nrows) lds[LDS_PAD (i)] = tr[u];
      }
   }
   for (int i = lane + 8 * NKP_WAVE; i < nrows; i += NKP_WAVE) lds[LDS_PAD (i)] = rhs[(int64_t) R0 + i];
   __syncthreads ();

   if (lane < nb) {
      const int s = s_pre;
      const int len = len_pre;
      const FT *ft = fl + lane;
      const int dstride = ml * gw;
      // the whole column lives in registers: no LDS write sits between two LDS reads, so the compiler
      // can keep the (read-only) factor reads in flight ahead of the dependent arithmetic
      double v[MAXL];
#pragma unroll
      for (int k = 0; k < MAXL; k++) v[k] = (k < len) ? lds[LDS_PAD (s + k)] : 0.0;
      // forward: y_k = ((r_k - l(k,k-P) y_{k-P}) ... - l(k,k-1) y_{k-1})   (far diagonal first, like the column sweep)
      // steps k >= len need no predicate: their factors are zero-padded, so they compute 0 - 0*x = 0;
      // the only branch left is wave-uniform (k < ml), which keeps the LDS reads hoistable
      // ml is a multiple of 8 (layout builder), so the only branch is one wave-uniform test per 8 steps and
      // the 8*P factor reads of a chunk are issued together, ahead of the dependent arithmetic
#pragma unroll
      for (int k0 = 0; k0 < MAXL; k0 += 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0; k < k0 + 8; k++) {
               double y = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * gw] * v[k - q];
               v[k] = y;
            }
         }
      }
      // backward: x_k = (((y_k - u(k,k+P) x_{k+P}) ... - u(k,k+1) x_{k+1}) * (1/u_kk)
#pragma unroll
      for (int k0 = MAXL - 8; k0 >= 0; k0 -= 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0 + 7; k >= k0; k--) {
               double x = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k + q < MAXL) x -= (double) ft[(P + q) * dstride + k * gw] * v[k + q];
               x *= (double) ft[P * dstride + k * gw];
               v[k] = x;
            }
         }
      }
#pragma unroll
      for (int k = 0; k < MAXL; k++)
         if (k < len) lds[LDS_PAD (s + k)] = v[k];
   }
   __syncthreads ();
   if (accumulate) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         if (i < nrows) z[(int64_t) R0 + i] = tz[u] + lds[LDS_PAD (i)];
      }
      for (int i = lane + 8 * NKP_WAVE; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] += lds[LDS_PAD (i)];
   } else
      for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] = lds[LDS_PAD (i)];
}

// The same kernel text once more for 80-level grids (65-80 rows per column), capped at three waves per SIMD: left alone
// the compiler spends all 256 VGPRs on hoisted factor reads and ONE wave per SIMD remains -- 0.25 degree x 80: 1120 us per
// colour of the fine level (1 TB/s); capped it spills 56 registers of a kernel in which 8 lanes compute and runs three
// waves per SIMD: 619 us, whole cycle 32.8 -> 24.9 ms, same bits.  The cap HURTS the 64-level kernel (3 degree: 16.1 -> 25.5
// us) and the streamed kernel (26 -> 94 us), so it is confined to this instantiation.  (A shared __device__ body would
// be tidier than a second copy of the text, but inlining it changed the register allocation of the other kernels.)
template <int P, int MAXL, class FT>
__global__ __launch_bounds__ (NKP_WAVE) __attribute__ ((amdgpu_waves_per_eu (3)))
void colblock_apply_lanes_kernel_w3 (const int *__restrict__ blk_start, const int *__restrict__ grp_b0, const int *__restrict__ grp_nb,
                                  const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                  const FT *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                  int gw, int rhs_slots, const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];            // [rhs_slots] staged right-hand side | [(2P+1)*ml*gw] the group's factors
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   // every index this wave needs depends on g alone: one round trip, not a chain through blk_start
   const int nb = grp_nb[g], ml = grp_maxlen[g];
   const int R0 = grp_row0[g], nrows = grp_row0[ngrp + g];
   int s_pre = 0, len_pre = 0;
   if (lane < gw) { s_pre = col_slot[g * gw + lane]; len_pre = col_slot[(ngrp + g) * gw + lane]; }
   FT *fl = reinterpret_cast<FT *> (lds + rhs_slots);
   // the accumulate target is requested together with the right-hand side
   double tz[8];
#pragma unroll
   for (int u = 0; u < 8; u++) {
      const int i = lane + u * NKP_WAVE;
      tz[u] = (accumulate && i < nrows) ? z[(int64_t) R0 + i] : 0.0;
   }
   // bulk, fully coalesced staging: every load is independent, so the whole group is in flight at once
   {
      // 16-byte units of the group's factor block (ml*gw is a multiple of 64, so this is exact for f32 too)
      const double2 *src = reinterpret_cast<const double2 *> (fac_t + grp_base[g]);
      double2 *dst = reinterpret_cast<double2 *> (fl);
      const int cnt2 = (int) (((size_t) (2 * P + 1) * ml * gw * sizeof (FT)) >> 4);
      // the right-hand side and a batch of 20 factor loads per lane are all in flight before the first LDS
      // store waits on them (a group of 8 columns x 64 levels x 5 diagonals is exactly one batch)
      double tr[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         tr[u] = (i < nrows) ? rhs[(int64_t) R0 + i] : 0.0;
      }
      constexpr int BATCH = sizeof (FT) == 4 ? 10 : 20;      // 16-byte loads per lane: one batch covers 8 x 64 x 5 factors
      for (int i0 = lane; i0 < cnt2; i0 += BATCH * NKP_WAVE) {
         double2 t[BATCH];
#pragma unroll
         for (int u = 0; u < BATCH; u++) {
            const int i = i0 + u * NKP_WAVE;
            t[u] = (i < cnt2) ? src[i] : make_double2 (0.0, 0.0);
         }
#pragma unroll
         for (int u = 0; u < BATCH; u++) {
            const int i = i0 + u * NKP_WAVE;
            if (i < cnt2) dst[i] = t[u];
         }
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         if (i < nrows) lds[LDS_PAD (i)] = tr[u];
      }
   }
   for (int i = lane + 8 * NKP_WAVE; i < nrows; i += NKP_WAVE) lds[LDS_PAD (i)] = rhs[(int64_t) R0 + i];
   __syncthreads ();

   if (lane < nb) {
      const int s = s_pre;
      const int len = len_pre;
      const FT *ft = fl + lane;
      const int dstride = ml * gw;
      // the whole column lives in registers: no LDS write sits between two LDS reads, so the compiler
      // can keep the (read-only) factor reads in flight ahead of the dependent arithmetic
      double v[MAXL];
#pragma unroll
      for (int k = 0; k < MAXL; k++) v[k] = (k < len) ? lds[LDS_PAD (s + k)] : 0.0;
      // forward: y_k = ((r_k - l(k,k-P) y_{k-P}) ... - l(k,k-1) y_{k-1})   (far diagonal first, like the column sweep)
      // steps k >= len need no predicate: their factors are zero-padded, so they compute 0 - 0*x = 0;
      // the only branch left is wave-uniform (k < ml), which keeps the LDS reads hoistable
      // ml is a multiple of 8 (layout builder), so the only branch is one wave-uniform test per 8 steps and
      // the 8*P factor reads of a chunk are issued together, ahead of the dependent arithmetic
#pragma unroll
      for (int k0 = 0; k0 < MAXL; k0 += 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0; k < k0 + 8; k++) {
               double y = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * gw] * v[k - q];
               v[k] = y;
            }
         }
      }
      // backward: x_k = (((y_k - u(k,k+P) x_{k+P}) ... - u(k,k+1) x_{k+1}) * (1/u_kk)
#pragma unroll
      for (int k0 = MAXL - 8; k0 >= 0; k0 -= 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0 + 7; k >= k0; k--) {
               double x = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k + q < MAXL) x -= (double) ft[(P + q) * dstride + k * gw] * v[k + q];
               x *= (double) ft[P * dstride + k * gw];
               v[k] = x;
            }
         }
      }
#pragma unroll
      for (int k = 0; k < MAXL; k++)
         if (k < len) lds[LDS_PAD (s + k)] = v[k];
   }
   __syncthreads ();
   if (accumulate) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         if (i < nrows) z[(int64_t) R0 + i] = tz[u] + lds[LDS_PAD (i)];
      }
      for (int i = lane + 8 * NKP_WAVE; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] += lds[LDS_PAD (i)];
   } else
      for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] = lds[LDS_PAD (i)];
}


// ================================================================ lane-per-column apply, 64 columns per wave, factors streamed
// The 8-columns-per-wave kernel above keeps 56 of 64 lanes idle during the substitution and stages the factors through
// LDS, which caps a CU at 11 waves; all of them load, then all of them compute, and HBM idles during the compute phase
// (1 degree fine level: 44 us per colour for 93 MB = 2.1 TB/s).  Here a wave owns GW = 32 (or 64) columns: only the
// right-hand side goes through LDS (coalesced staging of the group's contiguous rows), every lane below GW runs its own
// substitution, and each step reads its factors straight from HBM/L2 as ONE coalesced 128- or 256-byte load per diagonal
// ([diag][k][lane] layout) that the unrolled code requests many steps ahead (244 VGPRs: two waves per SIMD, all waves of
// a colour resident at once).  Same operations in the same order => same bits as the other kernels.
template <int P, int MAXL, class FT, int GW>
__global__ __launch_bounds__ (NKP_WAVE)
void colblock_apply_stream_kernel (const int *__restrict__ grp_nb, const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                   const FT *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                   const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];            // the group's right-hand side, then its solution
   constexpr int gw = GW;
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   const int nb = grp_nb[g], ml = grp_maxlen[g];
   const int R0 = grp_row0[g], nrows = grp_row0[ngrp + g];
   int s = 0, len = 0;
   if (lane < gw) { s = col_slot[g * gw + lane]; len = col_slot[(ngrp + g) * gw + lane]; }
   const FT *ft = fac_t + grp_base[g] + lane;
   const int dstride = ml * gw;
   for (int i0 = lane; i0 < nrows; i0 += 8 * NKP_WAVE) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = (i0 + u * NKP_WAVE < nrows) ? rhs[(int64_t) R0 + i0 + u * NKP_WAVE] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++)
         if (i0 + u * NKP_WAVE < nrows) lds[LDS_PAD (i0 + u * NKP_WAVE)] = t[u];
   }
   __syncthreads ();
   if (lane < nb) {
      double v[MAXL];
#pragma unroll
      for (int k = 0; k < MAXL; k++) v[k] = (k < len) ? lds[LDS_PAD (s + k)] : 0.0;
#pragma unroll
      for (int k0 = 0; k0 < MAXL; k0 += 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0; k < k0 + 8; k++) {
               double y = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * gw] * v[k - q];
               v[k] = y;
            }
         }
      }
#pragma unroll
      for (int k0 = MAXL - 8; k0 >= 0; k0 -= 8) {
         if (k0 < ml) {
#pragma unroll
            for (int k = k0 + 7; k >= k0; k--) {
               double x = v[k];
#pragma unroll
               for (int q = P; q >= 1; q--)
                  if (k + q < MAXL) x -= (double) ft[(P + q) * dstride + k * gw] * v[k + q];
               x *= (double) ft[P * dstride + k * gw];
               v[k] = x;
            }
         }
      }
#pragma unroll
      for (int k = 0; k < MAXL; k++)
         if (k < len) lds[LDS_PAD (s + k)] = v[k];
   }
   __syncthreads ();
   if (accumulate)
      for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] += lds[LDS_PAD (i)];
   else
      for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] = lds[LDS_PAD (i)];
}


// ================================================================ lane-per-column apply, 32 columns per wave, column resident in LDS
// The streamed kernel above keeps the whole column in registers (2 VGPRs per level): at 80 levels nothing is left for
// loads in flight and it loses to the small-group kernel, which in turn runs at a fifth of the HBM peak there.  Here the
// column stays in LDS, where the right-hand side is staged anyway: a substitution step reads its right-hand side from LDS
// and writes its result back, the registers hold only the P previous values and TWO chunks of CH steps' factors -- the
// chunk being consumed and the next one, requested before the current one is used -- so the register count does not
// depend on the column length.  Same operations in the same order as the other kernels => same bits.
template <int P, int CH, class FT>
struct FacChunk { FT f[P + 1][CH]; };

template <int P, int CH, class FT>
__device__ __forceinline__ void load_fwd (FacChunk<P, CH, FT> &c, const FT *__restrict__ ft, int dstride, int k0, int gw)
{
#pragma unroll
   for (int q = 1; q <= P; q++)
#pragma unroll
      for (int j = 0; j < CH; j++) c.f[q - 1][j] = ft[(P - q) * dstride + (k0 + j) * gw];
}

template <int P, int CH, class FT>
__device__ __forceinline__ void load_bwd (FacChunk<P, CH, FT> &c, const FT *__restrict__ ft, int dstride, int k0, int gw)
{
#pragma unroll
   for (int q = 0; q <= P; q++)
#pragma unroll
      for (int j = 0; j < CH; j++) c.f[q][j] = ft[(P + q) * dstride + (k0 + j) * gw];
}

// steps k0 .. k0 + CH - 1 of the forward substitution; w[q - 1] = y_{k - q} on entry and on exit
template <int P, int CH, class FT>
__device__ __forceinline__ void step_fwd (const FacChunk<P, CH, FT> &c, double *lds, int s, int len, int k0, double (&w)[P])
{
   double b[CH];
#pragma unroll
   for (int j = 0; j < CH; j++) b[j] = (k0 + j < len) ? lds[LDS_PAD (s + k0 + j)] : 0.0;
#pragma unroll
   for (int j = 0; j < CH; j++) {
      double y = b[j];
#pragma unroll
      for (int q = P; q >= 1; q--)
         if (k0 + j - q >= 0) y -= (double) c.f[q - 1][j] * w[q - 1];
#pragma unroll
      for (int q = P - 1; q >= 1; q--) w[q] = w[q - 1];
      w[0] = y;
      if (k0 + j < len) lds[LDS_PAD (s + k0 + j)] = y;
   }
}

// steps k0 + CH - 1 .. k0 of the back substitution; u[q - 1] = x_{k + q}
template <int P, int CH, class FT>
__device__ __forceinline__ void step_bwd (const FacChunk<P, CH, FT> &c, double *lds, int s, int len, int k0, double (&u)[P])
{
   double y[CH];
#pragma unroll
   for (int j = 0; j < CH; j++) y[j] = (k0 + j < len) ? lds[LDS_PAD (s + k0 + j)] : 0.0;
#pragma unroll
   for (int j = CH - 1; j >= 0; j--) {
      double x = y[j];
#pragma unroll
      for (int q = P; q >= 1; q--) x -= (double) c.f[q][j] * u[q - 1];
      x *= (double) c.f[0][j];
#pragma unroll
      for (int q = P - 1; q >= 1; q--) u[q] = u[q - 1];
      u[0] = x;
      if (k0 + j < len) lds[LDS_PAD (s + k0 + j)] = x;
   }
}

template <int P, class FT, int CH, bool EARLY>
__global__ __launch_bounds__ (NKP_WAVE)
void colblock_apply_ldsres_kernel (const int *__restrict__ grp_nb, const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                   const FT *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                   const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];            // the group's right-hand side, then its solution
   constexpr int gw = 32;
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   const int nb = grp_nb[g], ml = grp_maxlen[g];            // ml is a multiple of CH (layout built for this kernel)
   const int R0 = grp_row0[g], nrows = grp_row0[ngrp + g];
   int s = 0, len = 0;
   if (lane < gw) { s = col_slot[g * gw + lane]; len = col_slot[(ngrp + g) * gw + lane]; }
   const FT *ft = fac_t + grp_base[g] + lane;
   const int dstride = ml * gw;
   FacChunk<P, CH, FT> A, B, C;
   // EARLY: the first factor chunk of either sweep is requested before the right-hand side is staged, so the two round trips
   // overlap instead of following each other (the lanes of columns that do not exist read zero-padded factors)
   if (EARLY && lane < gw) {
      load_fwd<P, CH, FT> (A, ft, dstride, 0, gw);
      load_bwd<P, CH, FT> (C, ft, dstride, ml - CH, gw);
   }
   double tz[8];
   if (EARLY && accumulate) {
#pragma unroll
      for (int u = 0; u < 8; u++) tz[u] = (lane + u * NKP_WAVE < nrows) ? z[(int64_t) R0 + lane + u * NKP_WAVE] : 0.0;
   }
   for (int i0 = lane; i0 < nrows; i0 += 8 * NKP_WAVE) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = (i0 + u * NKP_WAVE < nrows) ? rhs[(int64_t) R0 + i0 + u * NKP_WAVE] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++)
         if (i0 + u * NKP_WAVE < nrows) lds[LDS_PAD (i0 + u * NKP_WAVE)] = t[u];
   }
   __syncthreads ();
   if (lane < nb) {
      double w[P];
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
      if (!EARLY) load_fwd<P, CH, FT> (A, ft, dstride, 0, gw);
      for (int k0 = 0; k0 < ml; k0 += 2 * CH) {
         const bool more = k0 + CH < ml;
         if (more) load_fwd<P, CH, FT> (B, ft, dstride, k0 + CH, gw);
         step_fwd<P, CH, FT> (A, lds, s, len, k0, w);
         if (more) {
            if (k0 + 2 * CH < ml) load_fwd<P, CH, FT> (A, ft, dstride, k0 + 2 * CH, gw);
            step_fwd<P, CH, FT> (B, lds, s, len, k0 + CH, w);
         }
      }
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
      if (!EARLY) load_bwd<P, CH, FT> (C, ft, dstride, ml - CH, gw);
      // the back substitution walks the chunks C, then B / A alternately
      int k0 = ml - CH;
      if (k0 - CH >= 0) load_bwd<P, CH, FT> (B, ft, dstride, k0 - CH, gw);
      step_bwd<P, CH, FT> (C, lds, s, len, k0, w);
      for (k0 -= CH; k0 >= 0; k0 -= 2 * CH) {
         const bool more = k0 - CH >= 0;
         if (more) load_bwd<P, CH, FT> (A, ft, dstride, k0 - CH, gw);
         step_bwd<P, CH, FT> (B, lds, s, len, k0, w);
         if (more) {
            if (k0 - 2 * CH >= 0) load_bwd<P, CH, FT> (B, ft, dstride, k0 - 2 * CH, gw);
            step_bwd<P, CH, FT> (A, lds, s, len, k0 - CH, w);
         }
      }
   }
   __syncthreads ();
   if (accumulate) {
      if (EARLY) {
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const int i = lane + u * NKP_WAVE;
            if (i < nrows) z[(int64_t) R0 + i] = tz[u] + lds[LDS_PAD (i)];
         }
         for (int i = lane + 8 * NKP_WAVE; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] += lds[LDS_PAD (i)];
      } else
         for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] += lds[LDS_PAD (i)];
   } else
      for (int i = lane; i < nrows; i += NKP_WAVE) z[(int64_t) R0 + i] = lds[LDS_PAD (i)];
}

// ================================================================ LDS-resident columns, factors packed along the column
// What held the kernel above at 0.43 of the HBM peak (1 degree fine level: 27 us for 93 MB): a colour is ONE round of
// waves (1468 groups of 32 columns on 256 CUs, 8 wave slots each), every wave walks its 2 x 4 factor chunks one behind
// the other with a single chunk of look-ahead, and a chunk is 32-48 load instructions of 128 bytes -- the 63 loads a
// wave may have in flight are 8 KB.  A latency-bound chain per wave with nothing to overlap it.
// Here the factors of 4 consecutive steps of a column sit side by side ([diag][k / 4][lane][4], f32), so one instruction
// moves 512 bytes and a chunk of 16 steps is 2-3 instructions per diagonal; and the schedule is static: every forward chunk
// is requested before the right-hand side is even staged, the backward chunks follow as the registers of consumed forward
// chunks come free (a compile-time budget of LDSP_BUDGET registers decides what is in flight when), so by the time the
// forward sweep is done the backward factors have landed.  Same operations in the same order => same bits.
#define LDSP_BUDGET 200

template <int NCH, int FR, int BR>
struct LdspSchedule {
   int f_upfront = 0, b_upfront = 0;
   int f_after[NCH] = {}, b_after_f[NCH] = {}, b_after_b[NCH] = {};     // chunks issued in total once fwd step c / bwd step j is done
   constexpr LdspSchedule ()
   {
      int fi = 0, bi = 0, fc = 0, bc = 0;
      while (fi < NCH && ((fi - fc) * FR + (bi - bc) * BR + FR <= LDSP_BUDGET || fi == fc)) fi++;
      while (bi < NCH && (fi - fc) * FR + (bi - bc) * BR + BR <= LDSP_BUDGET) bi++;
      f_upfront = fi;
      b_upfront = bi;
      for (int c = 0; c < NCH; c++) {
         fc++;
         while (fi < NCH && ((fi - fc) * FR + (bi - bc) * BR + FR <= LDSP_BUDGET || fi == fc)) fi++;
         while (bi < NCH && ((fi - fc) * FR + (bi - bc) * BR + BR <= LDSP_BUDGET || (fc == NCH && bi == bc))) bi++;
         f_after[c] = fi;
         b_after_f[c] = bi;
      }
      for (int j = 0; j < NCH; j++) {
         bc++;
         while (bi < NCH && ((bi - bc) * BR + BR <= LDSP_BUDGET || bi == bc)) bi++;
         b_after_b[j] = bi;
      }
   }
};

// a chunk of 16 steps as it comes off the loads: one float4 per diagonal and 4 steps
template <int P>
struct PackChunk { float4 q[P + 1][4]; };

// keeps a loaded float4 as it is until this point of the program: without it the compiler converts every factor to f64
// right behind its load (twice the registers, and a wait for the load where it was meant to stay in flight)
__device__ __forceinline__ void ldsp_pin (float4 &v) { asm volatile ("" : "+v" (v.x), "+v" (v.y), "+v" (v.z), "+v" (v.w)); }

template <int P>
__device__ __forceinline__ void ldsp_load_fwd (PackChunk<P> &c, const float4 *__restrict__ f4, int mlq, int k0, int gw)
{
#pragma unroll
   for (int q = 1; q <= P; q++)
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) c.q[q - 1][j4] = f4[((int64_t) (P - q) * mlq + (k0 >> 2) + j4) * gw];
}

template <int P>
__device__ __forceinline__ void ldsp_load_bwd (PackChunk<P> &c, const float4 *__restrict__ f4, int mlq, int k0, int gw)
{
#pragma unroll
   for (int q = 0; q <= P; q++)
#pragma unroll
      for (int j4 = 0; j4 < 4; j4++) c.q[q][j4] = f4[((int64_t) (P + q) * mlq + (k0 >> 2) + j4) * gw];
}

__device__ __forceinline__ float ldsp_elem (const float4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// steps k0 .. k0 + 15 of the forward substitution; the column sits at col[0 .. ml) in LDS (zero beyond its length)
template <int P>
__device__ __forceinline__ void ldsp_step_fwd (PackChunk<P> &c, double *col, int k0, double (&w)[P])
{
   double b[16];
#pragma unroll
   for (int j = 0; j < 16; j++) b[j] = col[k0 + j];
#pragma unroll
   for (int j4 = 0; j4 < 4; j4++) {
#pragma unroll
      for (int q = 0; q < P; q++) ldsp_pin (c.q[q][j4]);
#pragma unroll
      for (int i = 0; i < 4; i++) {
         const int j = 4 * j4 + i;
         double y = b[j];
#pragma unroll
         for (int q = P; q >= 1; q--) y -= (double) ldsp_elem (c.q[q - 1][j4], i) * w[q - 1];     // steps before the column's first: w = 0
#pragma unroll
         for (int q = P - 1; q >= 1; q--) w[q] = w[q - 1];
         w[0] = y;
         col[k0 + j] = y;
      }
   }
}

// steps k0 + 15 .. k0 of the back substitution
template <int P>
__device__ __forceinline__ void ldsp_step_bwd (PackChunk<P> &c, double *col, int k0, double (&u)[P])
{
   double y[16];
#pragma unroll
   for (int j = 0; j < 16; j++) y[j] = col[k0 + j];
#pragma unroll
   for (int j4 = 3; j4 >= 0; j4--) {
#pragma unroll
      for (int q = 0; q <= P; q++) ldsp_pin (c.q[q][j4]);
#pragma unroll
      for (int i = 3; i >= 0; i--) {
         const int j = 4 * j4 + i;
         double x = y[j];
#pragma unroll
         for (int q = P; q >= 1; q--) x -= (double) ldsp_elem (c.q[q][j4], i) * u[q - 1];
         x *= (double) ldsp_elem (c.q[0][j4], i);
#pragma unroll
         for (int q = P - 1; q >= 1; q--) u[q] = u[q - 1];
         u[0] = x;
         col[k0 + j] = x;
      }
   }
}

// LDS image: column l of the group at [l * LDSP_STRIDE, ...), LDSP_STRIDE = NCH * 16 + 1 doubles.  Odd stride: the 32 lanes of
// a substitution step hit 32 different bank pairs; fixed stride: every LDS address of the sweeps is "lane base + constant",
// no address arithmetic and no predicate (SQ counters of the first version: 21 VALU instructions per step, most of them
// the padded index of the linear image, and 41 % of a wave's life spent issuing).  Rows beyond a column's length are zero
// (their factors are zero too), so the sweeps need no length test.
template <int P, int NCH>
__global__ __launch_bounds__ (NKP_WAVE, 2)
void colblock_apply_ldspack_kernel (const int *__restrict__ grp_nb, const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                    const float *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                    const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];
   constexpr int gw = 32, CH = 16, STRIDE = NCH * CH + 1;
   constexpr LdspSchedule<NCH, P * CH, (P + 1) * CH> S;
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   const int ml = grp_maxlen[g];                            // a multiple of 16, at most NCH * 16
   const int nch = ml / CH, mlq = ml >> 2;
   const int R0 = grp_row0[g];
   const int cl = lane & (gw - 1);
   // first row (within the group) and length of lane's column; lanes 32-63 mirror 0-31 (the staging loops broadcast from them)
   const int s = col_slot[g * gw + cl], len = col_slot[(ngrp + g) * gw + cl];
   const float4 *f4 = reinterpret_cast<const float4 *> (fac_t + grp_base[g]) + cl;
   // The factor loads are unconditional so that the kernel is one straight line the static schedule can be written into: a
   // group with fewer than NCH chunks requests its last chunk again (an L2 hit) and skips the steps.
   // F[c] = forward chunk min (c, nch - 1);  Bq[j] = backward chunk max (nch - 1 - j, 0)
#define LDSP_FWD_K0(t) (((t) < nch ? (t) : nch - 1) * CH)
#define LDSP_BWD_K0(t) ((nch - 1 - (t) > 0 ? nch - 1 - (t) : 0) * CH)
   PackChunk<P> F[NCH], Bq[NCH];
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.f_upfront) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.b_upfront) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
   // staging, one column per step: 64 lanes read up to 64 consecutive rows of column c (coalesced) and write them to its slot,
   // zeros behind its end; 8 columns' loads are in flight together
#pragma unroll
   for (int c0 = 0; c0 < gw; c0 += 8) {
      double t[8], t2[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int sc = __builtin_amdgcn_readlane (s, c0 + u), lc = __builtin_amdgcn_readlane (len, c0 + u);
         t[u] = (lane < lc) ? rhs[(int64_t) R0 + sc + lane] : 0.0;
         if (NCH > 4) t2[u] = (lane + NKP_WAVE < lc) ? rhs[(int64_t) R0 + sc + lane + NKP_WAVE] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         lds[(c0 + u) * STRIDE + lane] = t[u];
         if (NCH > 4 && lane < STRIDE - 1 - NKP_WAVE) lds[(c0 + u) * STRIDE + lane + NKP_WAVE] = t2[u];
      }
   }
   __syncthreads ();
   if (lane < gw) {
      double *col = lds + lane * STRIDE;
      double w[P];
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
         if (c < nch) ldsp_step_fwd<P> (F[c], col, c * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.f_after[c - 1] : S.f_upfront) && t < S.f_after[c]) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.b_after_f[c - 1] : S.b_upfront) && t < S.b_after_f[c]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int j = 0; j < NCH; j++) {
         if (j < nch) ldsp_step_bwd<P> (Bq[j], col, (nch - 1 - j) * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (j ? S.b_after_b[j - 1] : S.b_after_f[NCH - 1]) && t < S.b_after_b[j]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
   }
#undef LDSP_FWD_K0
#undef LDSP_BWD_K0
   __syncthreads ();
   // back out, one column per step; the accumulate target of 8 columns is requested together
#pragma unroll
   for (int c0 = 0; c0 < gw; c0 += 8) {
      double t[8], t2[8];
      int sc[8], lc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         sc[u] = __builtin_amdgcn_readlane (s, c0 + u);
         lc[u] = __builtin_amdgcn_readlane (len, c0 + u);
         t[u] = (accumulate && lane < lc[u]) ? z[(int64_t) R0 + sc[u] + lane] : 0.0;
         if (NCH > 4) t2[u] = (accumulate && lane + NKP_WAVE < lc[u]) ? z[(int64_t) R0 + sc[u] + lane + NKP_WAVE] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         if (lane < lc[u]) z[(int64_t) R0 + sc[u] + lane] = t[u] + lds[(c0 + u) * STRIDE + lane];
         if (NCH > 4 && lane + NKP_WAVE < lc[u]) z[(int64_t) R0 + sc[u] + lane + NKP_WAVE] = t2[u] + lds[(c0 + u) * STRIDE + lane + NKP_WAVE];
      }
   }
}

// The same kernel on TWO right-hand sides of a K-interleaved batch (batch.hip): lanes 0-31 run the columns on system k0, lanes
// 32-63 the same columns on system k0 + 1 -- the half of the wave that idles in the single-vector kernel (and already loads
// the same factors) does the second system, so the factor stream is read once per pair.  rhs / z are K-interleaved
// (element (row, k) at row * K + k); K = 4 takes two launches (k0 = 0, 2).  Same operations per column => same bits.
template <int P, int NCH>
__global__ __launch_bounds__ (NKP_WAVE, 2)
void colblock_apply_ldspack2_kernel (const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                     const float *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                     const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp, int K, int k0)
{
   extern __shared__ double lds[];
   constexpr int gw = 32, CH = 16, STRIDE = NCH * CH + 1;
   constexpr LdspSchedule<NCH, P * CH, (P + 1) * CH> S;
   const int g = blockIdx.x + g_first;
   const int lane = threadIdx.x;
   const int ml = grp_maxlen[g];
   const int nch = ml / CH, mlq = ml >> 2;
   const int R0 = grp_row0[g];
   const int cl = lane & (gw - 1);
   const int s = col_slot[g * gw + cl], len = col_slot[(ngrp + g) * gw + cl];
   const float4 *f4 = reinterpret_cast<const float4 *> (fac_t + grp_base[g]) + cl;
#define LDSP_FWD_K0(t) (((t) < nch ? (t) : nch - 1) * CH)
#define LDSP_BWD_K0(t) ((nch - 1 - (t) > 0 ? nch - 1 - (t) : 0) * CH)
   PackChunk<P> F[NCH], Bq[NCH];
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.f_upfront) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.b_upfront) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
#pragma unroll
   for (int c0 = 0; c0 < gw; c0 += 8) {
      double2 t[8], t2[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int sc = __builtin_amdgcn_readlane (s, c0 + u), lc = __builtin_amdgcn_readlane (len, c0 + u);
         t[u] = (lane < lc) ? *reinterpret_cast<const double2 *> (rhs + ((int64_t) R0 + sc + lane) * K + k0) : make_double2 (0.0, 0.0);
         if (NCH > 4) t2[u] = (lane + NKP_WAVE < lc) ? *reinterpret_cast<const double2 *> (rhs + ((int64_t) R0 + sc + lane + NKP_WAVE) * K + k0) : make_double2 (0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         lds[(c0 + u) * STRIDE + lane] = t[u].x;
         lds[(gw + c0 + u) * STRIDE + lane] = t[u].y;
         if (NCH > 4 && lane < STRIDE - 1 - NKP_WAVE) {
            lds[(c0 + u) * STRIDE + lane + NKP_WAVE] = t2[u].x;
            lds[(gw + c0 + u) * STRIDE + lane + NKP_WAVE] = t2[u].y;
         }
      }
   }
   __syncthreads ();
   {
      double *col = lds + lane * STRIDE;                    // slot lane = (system half, column)
      double w[P];
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
         if (c < nch) ldsp_step_fwd<P> (F[c], col, c * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.f_after[c - 1] : S.f_upfront) && t < S.f_after[c]) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.b_after_f[c - 1] : S.b_upfront) && t < S.b_after_f[c]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int j = 0; j < NCH; j++) {
         if (j < nch) ldsp_step_bwd<P> (Bq[j], col, (nch - 1 - j) * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (j ? S.b_after_b[j - 1] : S.b_after_f[NCH - 1]) && t < S.b_after_b[j]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
   }
#undef LDSP_FWD_K0
#undef LDSP_BWD_K0
   __syncthreads ();
#pragma unroll
   for (int c0 = 0; c0 < gw; c0 += 8) {
      double2 t[8], t2[8];
      int sc[8], lc[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
         sc[u] = __builtin_amdgcn_readlane (s, c0 + u);
         lc[u] = __builtin_amdgcn_readlane (len, c0 + u);
         t[u] = (accumulate && lane < lc[u]) ? *reinterpret_cast<const double2 *> (z + ((int64_t) R0 + sc[u] + lane) * K + k0) : make_double2 (0.0, 0.0);
         if (NCH > 4) t2[u] = (accumulate && lane + NKP_WAVE < lc[u]) ? *reinterpret_cast<const double2 *> (z + ((int64_t) R0 + sc[u] + lane + NKP_WAVE) * K + k0) : make_double2 (0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
         if (lane < lc[u])
            *reinterpret_cast<double2 *> (z + ((int64_t) R0 + sc[u] + lane) * K + k0) = make_double2 (t[u].x + lds[(c0 + u) * STRIDE + lane], t[u].y + lds[(gw + c0 + u) * STRIDE + lane]);
         if (NCH > 4 && lane + NKP_WAVE < lc[u])
            *reinterpret_cast<double2 *> (z + ((int64_t) R0 + sc[u] + lane + NKP_WAVE) * K + k0) =
               make_double2 (t2[u].x + lds[(c0 + u) * STRIDE + lane + NKP_WAVE], t2[u].y + lds[(gw + c0 + u) * STRIDE + lane + NKP_WAVE]);
      }
   }
}

template <class T>
static int up (T **dst, const std::vector<T> &src, size_t *bytes)
{
   void *q = nullptr;
   const size_t b = (src.size () ? src.size () : 1) * sizeof (T);
   hipError_t e = hipMalloc (&q, b);
   if (e != hipSuccess) return (int) e;
   if (!src.empty ()) {
      e = hipMemcpy (q, src.data (), src.size () * sizeof (T), hipMemcpyHostToDevice);
      if (e != hipSuccess) { (void) hipFree (q); return (int) e; }
   }
   *dst = (T *) q;
   *bytes += b;
   return 0;
}

int colblock_build_lane_layout (ColBlocksDev &B, const int *h_blk_start, const int *ranges, int nranges,
                                int *grp_first, size_t *device_bytes, hipStream_t st, int f32, const int *h_rowptr)
{
   std::vector<int> b0, nb, ml, row0, nrow, cslot, clen, rb_ptr, rb_bd;
   bool gs_ok = h_rowptr != nullptr;
   std::vector<long long> base;
   // columns per wave: the group's factors + right-hand side must fit LDS several times per CU
   const nkp_tuning &T = B.tune ? *B.tune : nkp_builtin_tuning ();
   int gw = T.col_group;
   if (gw != 8 && gw != 16 && gw != 32 && gw != 64) gw = 8;
   // levels with many columns: 32 or 64 columns per wave, factors streamed from HBM instead of staged in LDS
   // (colblock_apply_stream_kernel); col_stream = 0 disables, col_stream_min = fewest columns of a level that uses it
   {
      // measured at 1 degree (93 MB per colour of the fine level): 8 columns per wave with LDS-staged factors 44.7 us,
      // 64 streamed 33.6 us, 32 streamed 26.2 us (3.55 TB/s); on levels below ~50 000 columns the fewer, longer waves of
      // the streamed kernel lose to the small-group kernel (whole cycle 2.69 -> 2.52 ms with the fine level only, 2.55 with
      // the first two levels, 2.61 with three)
      const int on = T.col_stream != 0, sgw = T.col_stream_gw == 64 ? 64 : 32;
      const bool env_min = T.col_stream_min >= 0;
      const int min_cols = env_min ? T.col_stream_min : 50000;
      // the column lives in registers: beyond 64 levels the kernel needs all 256 VGPRs (one wave per SIMD) and loses --
      // 0.25 degree x 80 levels: cycle 46.2 ms with it against 32.8 ms with the small-group kernel
      B.stream = on && ranges[nranges] - ranges[0] >= min_cols && B.max_len <= 64;
      if (B.stream) gw = sgw;
      // the column resident in LDS (colblock_apply_ldsres_kernel): the default from 20 000 columns (8000 when the columns
      // are longer than 64 levels, where the alternative runs at half the rate).  Same box, 1 degree, cycle time twice each:
      // LDS-resident on levels 0 and 1: 2.385 / 2.394 ms; streamed on level 0, small groups on level 1: 2.410 / 2.427;
      // streamed on level 0, LDS-resident on level 1: 2.434 / 2.409.  NKP_COL_LDSRES=1 keeps it to the long columns (and
      // the streamed kernel on the largest levels), =0 switches it off.
      const int lr = T.col_ldsres;
      const int ncols = ranges[nranges] - ranges[0];
      int min_long = env_min ? min_cols : 8000, min_short = env_min ? min_cols : 20000;
      if (T.col_ldsres_min > 0) min_long = min_short = T.col_ldsres_min;     // A/B: the LDS-resident kernels from that many columns only
      B.ldsres = lr > 0 && on && B.max_len <= 128 && ((B.max_len > 64 && ncols >= min_long) || (lr == 2 && ncols >= min_short));
      if (B.ldsres) { B.stream = 0; gw = 32; }
      // 2 = factors packed four steps to a 16-byte load (colblock_apply_ldspack_kernel): f32 storage, at most 5 chunks of 16
      // levels; the fused half sweep and the tail kernel read the plain layout
      if (B.ldsres && f32 && B.max_len <= 80 && T.col_ldsres_packed && !h_rowptr && T.ml_tail_rows <= 0) B.ldsres = 2;
   }
   while (!B.stream && !B.ldsres && gw > 8 && (size_t) ((2 * B.P + 2) * ((B.max_len + 7) & ~7) * gw) * sizeof (double) > 56 * 1024) gw >>= 1;
   B.gw = gw;
   long long total = 0;
   int lds_need = 0, fac_need = 0;
   const int ndiag = 2 * B.P + 1;
   // packed layout (colblock_apply_ldspack_kernel addresses every column through its own (first row, length) pair): the groups
   // of a colour are formed from its columns SORTED BY LENGTH, so that a group's zero padding (to its longest column, in chunks
   // of 16 steps) and its step count follow the columns it holds -- at 1 degree the factor stream shrinks from 1.23 to 1.09 of its
   // algorithmic size and three quarters of the waves run 48 instead of 64 steps
   std::vector<int> gcols;
   const bool sorted_groups = B.ldsres == 2 && T.col_sort_groups != 0;
   if (sorted_groups) {
      std::vector<int> order;
      for (int r = 0; r < nranges; r++) {
         grp_first[r] = (int) b0.size ();
         order.resize ((size_t) (ranges[r + 1] - ranges[r]));
         for (size_t q = 0; q < order.size (); q++) order[q] = ranges[r] + (int) q;
         std::stable_sort (order.begin (), order.end (), [&] (int a, int c) {
            return (h_blk_start[a + 1] - h_blk_start[a] + NKP_LDSRES_CH - 1) / NKP_LDSRES_CH > (h_blk_start[c + 1] - h_blk_start[c] + NKP_LDSRES_CH - 1) / NKP_LDSRES_CH;
         });
         for (size_t q0 = 0; q0 < order.size (); q0 += (size_t) gw) {
            const int cnt = (int) std::min ((size_t) gw, order.size () - q0);
            int m = 0;
            for (int c = 0; c < cnt; c++) m = std::max (m, h_blk_start[order[q0 + c] + 1] - h_blk_start[order[q0 + c]]);
            m = (m + NKP_LDSRES_CH - 1) / NKP_LDSRES_CH * NKP_LDSRES_CH;
            row0.push_back (0);                                 // slots are absolute first rows
            nrow.push_back (0);
            for (int c = 0; c < gw; c++) {
               const int col = c < cnt ? order[q0 + c] : -1;
               gcols.push_back (col);
               cslot.push_back (col >= 0 ? h_blk_start[col] : 0);
               clen.push_back (col >= 0 ? h_blk_start[col + 1] - h_blk_start[col] : 0);
            }
            b0.push_back (order[q0]);
            nb.push_back (cnt);
            ml.push_back (m);
            base.push_back (total);
            total += (long long) ndiag * m * gw;
         }
      }
   }
   for (int r = 0; r < nranges && !sorted_groups; r++) {
      grp_first[r] = (int) b0.size ();
      for (int b = ranges[r]; b < ranges[r + 1]; b += gw) {
         const int cnt = std::min (gw, ranges[r + 1] - b);
         int m = 0;
         for (int c = b; c < b + cnt; c++) m = std::max (m, h_blk_start[c + 1] - h_blk_start[c]);
         m = B.ldsres ? (m + NKP_LDSRES_CH - 1) / NKP_LDSRES_CH * NKP_LDSRES_CH : (m + 7) & ~7;   // the apply kernels step in chunks (zero-padded factors)
         const int rows = h_blk_start[b + cnt] - h_blk_start[b];
         lds_need = std::max (lds_need, LDS_PAD (rows) + 2);
         fac_need = std::max (fac_need, ndiag * m * gw);
         row0.push_back (h_blk_start[b]);
         nrow.push_back (rows);
         for (int c = 0; c < gw; c++) {
            cslot.push_back (c < cnt ? h_blk_start[b + c] - h_blk_start[b] : 0);
            clen.push_back (c < cnt ? h_blk_start[b + c + 1] - h_blk_start[b + c] : 0);
         }
         b0.push_back (b);
         nb.push_back (cnt);
         ml.push_back (m);
         base.push_back (total);
         total += (long long) ndiag * m * gw;
         if (gs_ok) {
            // row blocks of the fused sweep kernel: consecutive rows of the group, <= GS_NNZ entries and <= GS_THREADS rows each
            rb_ptr.push_back ((int) rb_bd.size ());
            int r = h_blk_start[b];
            const int rend = h_blk_start[b + cnt];
            while (r < rend) {
               int r2 = r + 1;
               if (h_rowptr[r2] - h_rowptr[r] > GS_NNZ) { gs_ok = false; break; }
               while (r2 < rend && r2 - r < GS_THREADS && h_rowptr[r2 + 1] - h_rowptr[r] <= GS_NNZ) r2++;
               rb_bd.push_back (r);
               r = r2;
            }
         }
      }
   }
   grp_first[nranges] = (int) b0.size ();
   if (gs_ok) {
      rb_ptr.push_back ((int) rb_bd.size ());
      rb_bd.push_back (ranges[nranges] > 0 ? h_blk_start[ranges[nranges]] : 0);
      int rc2;
      if ((rc2 = up (&B.gs_rb_ptr, rb_ptr, device_bytes)) || (rc2 = up (&B.gs_rb, rb_bd, device_bytes))) return rc2;
   }
   B.ngrp = (int) b0.size ();
   if (B.ldsres == 2) lds_need = std::max (lds_need, 32 * ((B.max_len <= 64 ? 4 : 5) * NKP_LDSRES_CH + 1));     // one fixed-stride slot per column
   lds_need = (lds_need + 1) & ~1;                 // keep the factor area 16-byte aligned
   B.rhs_slots = lds_need;
   if (!B.stream && !B.ldsres) lds_need += f32 ? (fac_need + 1) / 2 : fac_need;      // doubles
   B.lds_doubles = lds_need;
   int rc;
   if ((rc = up (&B.grp_b0, b0, device_bytes)) || (rc = up (&B.grp_nb, nb, device_bytes)) || (rc = up (&B.grp_maxlen, ml, device_bytes)) ||
       (rc = up (&B.grp_base, base, device_bytes)))
      return rc;
   row0.insert (row0.end (), nrow.begin (), nrow.end ());
   cslot.insert (cslot.end (), clen.begin (), clen.end ());
   if ((rc = up (&B.grp_row0, row0, device_bytes)) || (rc = up (&B.col_slot, cslot, device_bytes))) return rc;
   void *q = nullptr;
   const size_t fsz = f32 ? sizeof (float) : sizeof (double);
   hipError_t e = hipMalloc (&q, (size_t) (total ? total : 1) * fsz);
   if (e != hipSuccess) return (int) e;
   if (f32) B.fac_tf = (float *) q;
   else B.fac_t = (double *) q;
   *device_bytes += (size_t) total * fsz;
   int *d_gcols = nullptr;
   if (sorted_groups && (rc = up (&d_gcols, gcols, device_bytes))) return rc;
   if (B.ngrp)
      hipLaunchKernelGGL (colblock_transpose_kernel, dim3 (B.ngrp), dim3 (NKP_WAVE), 0, st, B.blk_start, B.grp_b0, B.grp_nb, B.grp_maxlen,
                          B.grp_base, ndiag, B.n, B.fac, B.fac_t, gw, B.fac_tf, B.ldsres == 2 ? 4 : 1, (const int *) d_gcols);
   if (d_gcols) {
      (void) hipStreamSynchronize (st);
      (void) hipFree (d_gcols);
      *device_bytes -= gcols.size () * sizeof (int);
   }
   if (gs_ok) {
      const int gs_bytes = (GS_NNZ + lds_need) * (int) sizeof (double);
      if (gs_bytes > 64 * 1024) gs_ok = false;          // not worth running one workgroup per CU
      B.gs_lds_bytes = gs_bytes;
   }
   B.gs_ok = gs_ok ? 1 : 0;
   // dynamic LDS above the default limit needs an explicit opt-in
   const int lds_bytes = lds_need * (int) sizeof (double);
   if (lds_bytes > 160 * 1024) return (int) hipErrorInvalidValue;
   if (lds_bytes > 48 * 1024) {
#define LDS_OPT_IN(PP, ML)                                                                                                              \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_lanes_kernel<PP, ML, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_lanes_kernel<PP, ML, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
      LDS_OPT_IN (1, 64); LDS_OPT_IN (2, 64); LDS_OPT_IN (4, 64); LDS_OPT_IN (1, 128); LDS_OPT_IN (2, 128); LDS_OPT_IN (4, 128);
#undef LDS_OPT_IN
#define LDS_OPT_IN(PP, ML)                                                                                                              \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_lanes_kernel_w3<PP, ML, double>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_lanes_kernel_w3<PP, ML, float>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
      LDS_OPT_IN (1, 80); LDS_OPT_IN (2, 80); LDS_OPT_IN (4, 80);
#undef LDS_OPT_IN
#define LDS_OPT_IN(PP, ML)                                                                                                              \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_stream_kernel<PP, ML, double, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_stream_kernel<PP, ML, float, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);  \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_stream_kernel<PP, ML, double, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_stream_kernel<PP, ML, float, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
      LDS_OPT_IN (1, 64); LDS_OPT_IN (2, 64); LDS_OPT_IN (4, 64); LDS_OPT_IN (1, 96); LDS_OPT_IN (2, 96); LDS_OPT_IN (4, 96);
#undef LDS_OPT_IN
#define LDS_OPT_IN(PP)                                                                                                                    \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldsres_kernel<PP, double, NKP_LDSRES_CH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldsres_kernel<PP, float, NKP_LDSRES_CH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);  \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldsres_kernel<PP, double, NKP_LDSRES_CH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldsres_kernel<PP, float, NKP_LDSRES_CH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
      LDS_OPT_IN (1); LDS_OPT_IN (2); LDS_OPT_IN (4);
#undef LDS_OPT_IN
#define LDS_OPT_IN(PP)                                                                                                                    \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldspack_kernel<PP, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); \
      (void) hipFuncSetAttribute ((const void *) colblock_apply_ldspack_kernel<PP, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
      LDS_OPT_IN (1); LDS_OPT_IN (2); LDS_OPT_IN (4);
#undef LDS_OPT_IN
   }
   return (int) hipStreamSynchronize (st);
}

// Software-pipelined variant for the common shape (f32 factors, columns of at most 64 levels, 8 columns per wave,
// half bandwidth <= 2).  SQ counters on the one-group-per-wave kernel above: a wave lives ~14 us, 46 % of it parked on
// the staging loads, and LDS (14.5 KB per wave) caps a CU at 11 waves.  Here a persistent wave walks groups g,
// g + gridDim.x, ...: it commits the prefetched registers of group i to LDS, requests group i+1 (index words, then
// right-hand side, accumulate target and the factor block: 26 registers of loads in flight) and only then runs group
// i's recurrence, so the HBM latency hides under arithmetic.  Same arithmetic in the same order => same bits.
#define LANES_PIPE_BATCH 10
template <int P>
__global__ __launch_bounds__ (NKP_WAVE)
void colblock_apply_lanes_pipe_kernel (const int *__restrict__ grp_nb, const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base,
                                       int g_first, int g_end, const float *__restrict__ fac_t, const double *__restrict__ rhs,
                                       double *__restrict__ z, int accumulate, int rhs_slots, const int *__restrict__ grp_row0,
                                       const int *__restrict__ col_slot, int ngrp)
{
   extern __shared__ double lds[];
   constexpr int gw = 8, MAXL = 64;
   float *fl = reinterpret_cast<float *> (lds + rhs_slots);
   const int lane = threadIdx.x;
   int g = g_first + (int) blockIdx.x;
   if (g >= g_end) return;

   int n_nb, n_ml, n_R0, n_nrows, n_s, n_len;
   double n_tz[8], n_tr[8];
   double2 n_t[LANES_PIPE_BATCH];
#define LANES_PREFETCH(GG)                                                                            \
   do {                                                                                               \
      n_nb = grp_nb[GG];                                                                              \
      n_ml = grp_maxlen[GG];                                                                          \
      n_R0 = grp_row0[GG];                                                                            \
      n_nrows = grp_row0[ngrp + (GG)];                                                                \
      n_s = n_len = 0;                                                                                \
      if (lane < gw) { n_s = col_slot[(GG) * gw + lane]; n_len = col_slot[(ngrp + (GG)) * gw + lane]; } \
      _Pragma ("unroll")                                                                              \
      for (int u = 0; u < 8; u++) {                                                                   \
         const int i = lane + u * NKP_WAVE;                                                           \
         n_tz[u] = (accumulate && i < n_nrows) ? z[(int64_t) n_R0 + i] : 0.0;                         \
         n_tr[u] = (i < n_nrows) ? rhs[(int64_t) n_R0 + i] : 0.0;                                     \
      }                                                                                               \
      {                                                                                               \
         const double2 *src_ = reinterpret_cast<const double2 *> (fac_t + grp_base[GG]);              \
         const int cnt2_ = (int) (((size_t) (2 * P + 1) * n_ml * gw * sizeof (float)) >> 4);          \
         _Pragma ("unroll")                                                                           \
         for (int u = 0; u < LANES_PIPE_BATCH; u++) {                                                 \
            const int i = lane + u * NKP_WAVE;                                                        \
            n_t[u] = (i < cnt2_) ? src_[i] : make_double2 (0.0, 0.0);                                 \
         }                                                                                            \
      }                                                                                               \
   } while (0)

   LANES_PREFETCH (g);
   for (;;) {
      // commit the prefetched group to LDS; it becomes the current one
      const int nb = n_nb, ml = n_ml, R0 = n_R0, nrows = n_nrows, s = n_s, len = n_len;
      double tz[8];
      {
         double2 *dst = reinterpret_cast<double2 *> (fl);
         const int cnt2 = (int) (((size_t) (2 * P + 1) * ml * gw * sizeof (float)) >> 4);
#pragma unroll
         for (int u = 0; u < LANES_PIPE_BATCH; u++) {
            const int i = lane + u * NKP_WAVE;
            if (i < cnt2) dst[i] = n_t[u];
         }
#pragma unroll
         for (int u = 0; u < 8; u++) {
            const int i = lane + u * NKP_WAVE;
            tz[u] = n_tz[u];
            if (i < nrows) lds[LDS_PAD (i)] = n_tr[u];
         }
      }
      __syncthreads ();
      const int gn = g + (int) gridDim.x;
      const bool have_next = gn < g_end;
      if (have_next) LANES_PREFETCH (gn);

      if (lane < nb) {
         const float *ft = fl + lane;
         const int dstride = ml * gw;
         double v[MAXL];
#pragma unroll
         for (int k = 0; k < MAXL; k++) v[k] = (k < len) ? lds[LDS_PAD (s + k)] : 0.0;
#pragma unroll
         for (int k0 = 0; k0 < MAXL; k0 += 8) {
            if (k0 < ml) {
#pragma unroll
               for (int k = k0; k < k0 + 8; k++) {
                  double y = v[k];
#pragma unroll
                  for (int q = P; q >= 1; q--)
                     if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * gw] * v[k - q];
                  v[k] = y;
               }
            }
         }
#pragma unroll
         for (int k0 = MAXL - 8; k0 >= 0; k0 -= 8) {
            if (k0 < ml) {
#pragma unroll
               for (int k = k0 + 7; k >= k0; k--) {
                  double x = v[k];
#pragma unroll
                  for (int q = P; q >= 1; q--)
                     if (k + q < MAXL) x -= (double) ft[(P + q) * dstride + k * gw] * v[k + q];
                  x *= (double) ft[P * dstride + k * gw];
                  v[k] = x;
               }
            }
         }
#pragma unroll
         for (int k = 0; k < MAXL; k++)
            if (k < len) lds[LDS_PAD (s + k)] = v[k];
      }
      __syncthreads ();
#pragma unroll
      for (int u = 0; u < 8; u++) {
         const int i = lane + u * NKP_WAVE;
         if (i < nrows) z[(int64_t) R0 + i] = accumulate ? tz[u] + lds[LDS_PAD (i)] : lds[LDS_PAD (i)];
      }
      __syncthreads ();          // the LDS image is free again
      if (!have_next) break;
      g = gn;
   }
#undef LANES_PREFETCH
}


// ================================================================ fused Gauss-Seidel half sweep
// One launch per colour instead of two (residual SpMV over the colour's rows, then the column solves): a workgroup owns
// one group of <= gw consecutive water columns, computes r = b - L x for the group's rows block by block in the SpMV's
// CSR-stream fashion (coalesced (value, column) streams, x gathered through L2, products parked in LDS, one lane per row
// sums its segment) straight into LDS, then lanes 0..nb-1 of its first wave run the band substitutions on the LDS image,
// and the group's rows of x are written once.  r never goes to HBM, the factor block is requested before the first row
// block so its latency hides under the SpMV phase, and half of the cycle's launches disappear (the small levels sit at
// their launch-latency floor).
// x comes from TWO buffers: rows < split (colour 0) from xa, the rest from xb, and the new values of this colour go to
// xout -- the columns of one colour are coupled to each other (upwind3's +-2 neighbours, stub columns), so updating x in
// place would make the result depend on which workgroup ran first.  The caller ping-pongs the buffers (multilevel.hip).
// Same products, same per-row summation order, same substitution order as the two-kernel path => identical bits.
template <int P, class FT, class VT>
__global__ __launch_bounds__ (GS_THREADS)
void gs_fused_kernel (const int *__restrict__ rowptr, const int *__restrict__ colind, const VT *__restrict__ val,
                      const int *__restrict__ grp_nb, const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                      const FT *__restrict__ fac_t, int gw, int rhs_slots, const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp,
                      const int *__restrict__ gs_rb_ptr, const int *__restrict__ gs_rb,
                      const double *__restrict__ xa, const double *__restrict__ xb, int split, const double *__restrict__ b, double *__restrict__ xout)
{
   extern __shared__ double lds[];            // prod[GS_NNZ] | rs[rhs_slots] | the group's factors
   double *prod = lds;
   double *rs = lds + GS_NNZ;
   FT *fl = reinterpret_cast<FT *> (rs + rhs_slots);
   const int g = blockIdx.x + g_first;
   const int tid = threadIdx.x;
   const int nb = grp_nb[g], ml = grp_maxlen[g];
   const int R0 = grp_row0[g], nrows = grp_row0[ngrp + g];
   const int rb0 = gs_rb_ptr[g], rb1 = gs_rb_ptr[g + 1];
   int s_pre = 0, len_pre = 0;
   if (tid < gw) { s_pre = col_slot[g * gw + tid]; len_pre = col_slot[(ngrp + g) * gw + tid]; }
   // the factor block: requested now, parked in registers, committed to LDS after the SpMV phase
   constexpr int FB = 4;                       // 16-byte loads per thread and batch
   const double2 *fsrc = reinterpret_cast<const double2 *> (fac_t + grp_base[g]);
   double2 *fdst = reinterpret_cast<double2 *> (fl);
   const int cnt2 = (int) (((size_t) (2 * P + 1) * ml * gw * sizeof (FT)) >> 4);
   double2 ft0[FB];
#pragma unroll
   for (int u = 0; u < FB; u++) {
      const int i = tid + u * GS_THREADS;
      ft0[u] = (i < cnt2) ? fsrc[i] : make_double2 (0.0, 0.0);
   }
   // residual of the group's rows, row block by row block
   for (int rb = rb0; rb < rb1; rb++) {
      const int r0 = gs_rb[rb], r1 = gs_rb[rb + 1];
      const int e0 = rowptr[r0], e1 = rowptr[r1];
      const int cnt = e1 - e0;
      int seg0 = 0, seg1 = 0;
      double bv = 0.0;
      if (r0 + tid < r1) { seg0 = rowptr[r0 + tid]; seg1 = rowptr[r0 + tid + 1]; bv = b[r0 + tid]; }
#pragma unroll 4
      for (int k = tid; k < cnt; k += GS_THREADS) {
         const int c = colind[e0 + k];
         const double xv = (c < split) ? xa[c] : xb[c];
         prod[k] = (double) val[e0 + k] * xv;
      }
      __syncthreads ();
      if (r0 + tid < r1) {
         double acc = 0.0;
         const int s0 = seg0 - e0, s1 = seg1 - e0;
#pragma unroll 4
         for (int k = s0; k < s1; k++) acc += prod[k];
         rs[LDS_PAD (r0 + tid - R0)] = bv - acc;
      }
      __syncthreads ();
   }
   // factors into LDS (first batch from the registers, any rest straight through)
#pragma unroll
   for (int u = 0; u < FB; u++) {
      const int i = tid + u * GS_THREADS;
      if (i < cnt2) fdst[i] = ft0[u];
   }
   for (int i = tid + FB * GS_THREADS; i < cnt2; i += GS_THREADS) fdst[i] = fsrc[i];
   __syncthreads ();

   if (tid < nb) {
      const int s = s_pre, len = len_pre;
      const FT *ft = fl + tid;
      const int dstride = ml * gw;
      // forward: y_k = ((r_k - l(k,k-P) y_{k-P}) ... - l(k,k-1) y_{k-1}), 8 steps per LDS round trip
      double carry[P];
#pragma unroll
      for (int q = 0; q < P; q++) carry[q] = 0.0;            // carry[q-1] = y_{k0-q}
      for (int k0 = 0; k0 < ml; k0 += 8) {
         double t[8];
#pragma unroll
         for (int j = 0; j < 8; j++) t[j] = (k0 + j < len) ? rs[LDS_PAD (s + k0 + j)] : 0.0;
#pragma unroll
         for (int j = 0; j < 8; j++) {
            const int k = k0 + j;
            double y = t[j];
#pragma unroll
            for (int q = P; q >= 1; q--) {
               const double prev = (j - q >= 0) ? t[j - q >= 0 ? j - q : 0] : carry[q - j - 1 >= 0 && q - j - 1 < P ? q - j - 1 : 0];
               if (k - q >= 0) y -= (double) ft[(P - q) * dstride + k * gw] * prev;
            }
            t[j] = y;
         }
#pragma unroll
         for (int j = 0; j < 8; j++)
            if (k0 + j < len) rs[LDS_PAD (s + k0 + j)] = t[j];
#pragma unroll
         for (int q = 1; q <= P; q++) carry[q - 1] = t[8 - q];
      }
      // backward: x_k = (((y_k - u(k,k+P) x_{k+P}) ... - u(k,k+1) x_{k+1}) * (1/u_kk)
      double nxt[P];
#pragma unroll
      for (int q = 0; q < P; q++) nxt[q] = 0.0;              // nxt[q] = x_{k0+8+q}
      for (int k0 = ml - 8; k0 >= 0; k0 -= 8) {
         double t[8];
#pragma unroll
         for (int j = 0; j < 8; j++) t[j] = (k0 + j < len) ? rs[LDS_PAD (s + k0 + j)] : 0.0;
#pragma unroll
         for (int j = 7; j >= 0; j--) {
            const int k = k0 + j;
            double x = t[j];
#pragma unroll
            for (int q = P; q >= 1; q--) {
               const double nv = (j + q <= 7) ? t[j + q <= 7 ? j + q : 7] : nxt[j + q - 8 >= 0 && j + q - 8 < P ? j + q - 8 : 0];
               if (k + q < ml) x -= (double) ft[(P + q) * dstride + k * gw] * nv;
            }
            x *= (double) ft[P * dstride + k * gw];
            t[j] = x;
         }
#pragma unroll
         for (int j = 0; j < 8; j++)
            if (k0 + j < len) rs[LDS_PAD (s + k0 + j)] = t[j];
#pragma unroll
         for (int q = 0; q < P; q++) nxt[q] = t[q];
      }
   }
   __syncthreads ();
   for (int i = tid; i < nrows; i += GS_THREADS) {
      const int row = R0 + i;
      const double xo = (row < split) ? xa[row] : xb[row];
      xout[row] = xo + rs[LDS_PAD (i)];
   }
}

int launch_gs_fused (const CsrDev &L, const ColBlocksDev &B, int g0, int g1, const double *xa, const double *xb, int split, const double *b, double *xout, hipStream_t st)
{
   if (!B.gs_ok) return 1;
   if (g1 <= g0) return 0;
   static bool opted = false;
   if (!opted) {
      opted = true;
#define GS_OPT_IN(PP)                                                                                                                    \
      (void) hipFuncSetAttribute ((const void *) gs_fused_kernel<PP, float, float>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);   \
      (void) hipFuncSetAttribute ((const void *) gs_fused_kernel<PP, double, double>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)
      GS_OPT_IN (1); GS_OPT_IN (2); GS_OPT_IN (4);
#undef GS_OPT_IN
   }
#define GS_LAUNCH(PP)                                                                                                                                       \
   do {                                                                                                                                                     \
      if (B.fac_tf && L.valf)                                                                                                                               \
         hipLaunchKernelGGL ((gs_fused_kernel<PP, float, float>), dim3 (g1 - g0), dim3 (GS_THREADS), (size_t) B.gs_lds_bytes, st, L.rowptr, L.colind, L.valf, \
                             B.grp_nb, B.grp_maxlen, B.grp_base, g0, B.fac_tf, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp, B.gs_rb_ptr, B.gs_rb,         \
                             xa, xb, split, b, xout);                                                                                                        \
      else if (B.fac_t && !L.valf)                                                                                                                          \
         hipLaunchKernelGGL ((gs_fused_kernel<PP, double, double>), dim3 (g1 - g0), dim3 (GS_THREADS), (size_t) B.gs_lds_bytes, st, L.rowptr, L.colind, L.val, \
                             B.grp_nb, B.grp_maxlen, B.grp_base, g0, B.fac_t, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp, B.gs_rb_ptr, B.gs_rb,           \
                             xa, xb, split, b, xout);                                                                                                        \
      else return 1;                                                                                                                                        \
   } while (0)
   if (B.P == 1) GS_LAUNCH (1);
   else if (B.P == 2) GS_LAUNCH (2);
   else GS_LAUNCH (4);
#undef GS_LAUNCH
   return 0;
}

void launch_colblock_apply_lanes (const ColBlocksDev &B, int g0, int g1, const double *r, double *z, int accumulate, hipStream_t st)
{
   if (g1 <= g0) return;
   const size_t lds = (size_t) B.lds_doubles * sizeof (double);
   {
      // pipelined persistent variant: f32 factors, <= 64 levels, 8 columns per wave, band <= 2.  OFF unless
      // NKP_COLPIPE_MIN=<groups> is set: it needs 256 VGPRs (one wave per SIMD), and with nothing to interleave the
      // recurrence's own dependency stalls cost more than the hidden load latency saves -- 1 degree V-cycle 3.10 ms
      // against 2.70 ms for the one-group-per-wave kernel (bit-identical results)
      const int pipe_min = B.tune ? B.tune->col_pipe_min : 0;
      if (B.fac_tf && B.max_len <= 64 && B.gw == 8 && B.P <= 2 && pipe_min > 0 && g1 - g0 >= pipe_min && lds <= 48 * 1024) {
         int waves = 256 * 8;                         // two waves per SIMD fit the ~230 registers
         if (waves > (g1 - g0 + 1) / 2) waves = (g1 - g0 + 1) / 2;
         if (B.P == 1) hipLaunchKernelGGL ((colblock_apply_lanes_pipe_kernel<1>), dim3 (waves), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen, B.grp_base, g0, g1,
                                           B.fac_tf, r, z, accumulate, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);
         else hipLaunchKernelGGL ((colblock_apply_lanes_pipe_kernel<2>), dim3 (waves), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen, B.grp_base, g0, g1,
                                  B.fac_tf, r, z, accumulate, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);
         return;
      }
   }
   if (B.ldsres == 2) {
#define LDSP_LAUNCH(PP) do { if (B.max_len <= 64) hipLaunchKernelGGL ((colblock_apply_ldspack_kernel<PP, 4>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen, \
                                                                        B.grp_base, g0, B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp);                                \
                             else hipLaunchKernelGGL ((colblock_apply_ldspack_kernel<PP, 5>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen,                      \
                                                      B.grp_base, g0, B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp); } while (0)
      if (B.P == 1) LDSP_LAUNCH (1);
      else if (B.P == 2) LDSP_LAUNCH (2);
      else LDSP_LAUNCH (4);
#undef LDSP_LAUNCH
      return;
   }
   if (B.ldsres) {
#define LDSRES_LAUNCH2(PP, EE)                                                                                                                                    \
      do {                                                                                                                                                       \
         if (B.fac_tf) hipLaunchKernelGGL ((colblock_apply_ldsres_kernel<PP, float, NKP_LDSRES_CH, EE>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen, \
                                           B.grp_base, g0, B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp);                                           \
         else hipLaunchKernelGGL ((colblock_apply_ldsres_kernel<PP, double, NKP_LDSRES_CH, EE>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen,         \
                                  B.grp_base, g0, B.fac_t, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp);                                                    \
      } while (0)
      // col_ldsres_early = 1: first factor chunks and the accumulate target requested before the right-hand side is staged
      // (247 instead of 172 VGPRs).  Measured twice on one box: 26.9 / 26.6 us per colour of the 1 degree fine level with it,
      // 26.8 / 27.0 without -- no difference, so it stays off
      const int early = B.tune ? B.tune->col_ldsres_early : 0;
#define LDSRES_LAUNCH(PP) do { if (early) LDSRES_LAUNCH2 (PP, true); else LDSRES_LAUNCH2 (PP, false); } while (0)
      if (B.P == 1) LDSRES_LAUNCH (1);
      else if (B.P == 2) LDSRES_LAUNCH (2);
      else LDSRES_LAUNCH (4);
#undef LDSRES_LAUNCH
#undef LDSRES_LAUNCH2
      return;
   }
   if (B.stream) {
#define STREAM_LAUNCH3(PP, ML, GG)                                                                                                                              \
      do {                                                                                                                                                     \
         if (B.fac_tf) hipLaunchKernelGGL ((colblock_apply_stream_kernel<PP, ML, float, GG>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen,  \
                                           B.grp_base, g0, B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp);                                         \
         else hipLaunchKernelGGL ((colblock_apply_stream_kernel<PP, ML, double, GG>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_nb, B.grp_maxlen,          \
                                  B.grp_base, g0, B.fac_t, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp);                                                  \
      } while (0)
#define STREAM_LAUNCH2(PP, ML) do { if (B.gw == 32) STREAM_LAUNCH3 (PP, ML, 32); else STREAM_LAUNCH3 (PP, ML, 64); } while (0)
#define STREAM_LAUNCH(PP) do { if (B.max_len <= 64) STREAM_LAUNCH2 (PP, 64); else STREAM_LAUNCH2 (PP, 96); } while (0)
      if (B.P == 1) STREAM_LAUNCH (1);
      else if (B.P == 2) STREAM_LAUNCH (2);
      else STREAM_LAUNCH (4);
#undef STREAM_LAUNCH
#undef STREAM_LAUNCH2
#undef STREAM_LAUNCH3
      return;
   }
   const int use_w3 = B.tune ? B.tune->col_w3 : 1;
#define LANES_LAUNCH_W3(PP, ML)                                                                                                                                 \
   do {                                                                                                                                                        \
      if (B.fac_tf) hipLaunchKernelGGL ((colblock_apply_lanes_kernel_w3<PP, ML, float>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.blk_start, B.grp_b0, B.grp_nb, \
                                        B.grp_maxlen, B.grp_base, g0, B.fac_tf, r, z, accumulate, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);             \
      else hipLaunchKernelGGL ((colblock_apply_lanes_kernel_w3<PP, ML, double>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.blk_start, B.grp_b0, B.grp_nb,          \
                               B.grp_maxlen, B.grp_base, g0, B.fac_t, r, z, accumulate, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);                       \
   } while (0)
#define LANES_LAUNCH(PP)                                                                                                   \
   do {                                                                                                                    \
      if (B.max_len <= 64) LANES_LAUNCH2 (PP, 64);                                                                         \
      else if (B.max_len <= 80 && use_w3) LANES_LAUNCH_W3 (PP, 80);                                                        \
      else LANES_LAUNCH2 (PP, 128);                                                                                        \
   } while (0)
#define LANES_LAUNCH2(PP, ML)                                                                                                                                   \
   do {                                                                                                                                                        \
      if (B.fac_tf) hipLaunchKernelGGL ((colblock_apply_lanes_kernel<PP, ML, float>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.blk_start, B.grp_b0, B.grp_nb, \
                                        B.grp_maxlen, B.grp_base, g0, B.fac_tf, r, z, accumulate, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);             \
      else hipLaunchKernelGGL ((colblock_apply_lanes_kernel<PP, ML, double>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.blk_start, B.grp_b0, B.grp_nb,          \
                               B.grp_maxlen, B.grp_base, g0, B.fac_t, r, z, accumulate, B.gw, B.rhs_slots, B.grp_row0, B.col_slot, B.ngrp);                       \
   } while (0)
   if (B.P == 1) LANES_LAUNCH (1);
   else if (B.P == 2) LANES_LAUNCH (2);
   else LANES_LAUNCH (4);
#undef LANES_LAUNCH
#undef LANES_LAUNCH2
#undef LANES_LAUNCH_W3
}

// FOUR right-hand sides in one launch: a wave takes 16 of a group's 32 columns (block 2 g + h = half h of group g) for all four
// systems -- lane = system * 16 + column.  The factors are read once for the four systems (the 16 lanes of every system address
// the same 256 bytes), the staging moves whole 32-byte rows of the interleaved vectors.  The two-system kernel above takes two
// launches for K = 4 and its 33 KB of LDS per wave leave a colour's 1469 waves in 1.4 rounds of 4 per CU: 53 us per launch
// at 1 degree, slower per system than the single-vector kernel; here the same LDS holds 4 x 16 slots and the waves are twice as many.
template <int P, int NCH>
__global__ __launch_bounds__ (NKP_WAVE, 2)
void colblock_apply_ldspack4_kernel (const int *__restrict__ grp_maxlen, const long long *__restrict__ grp_base, int g_first,
                                     const float *__restrict__ fac_t, const double *__restrict__ rhs, double *__restrict__ z, int accumulate,
                                     const int *__restrict__ grp_row0, const int *__restrict__ col_slot, int ngrp, int K /* vectors interleaved */, int k0 /* first of this launch's four */)
{
   extern __shared__ double lds[];
   constexpr int gw = 32, hw = 16, CH = 16, STRIDE = NCH * CH + 1;
   rhs += k0;
   z += k0;
   constexpr LdspSchedule<NCH, P * CH, (P + 1) * CH> S;
   const int g = (int) (blockIdx.x >> 1) + g_first, half = blockIdx.x & 1;
   const int lane = threadIdx.x;
   const int ml = grp_maxlen[g];
   const int nch = ml / CH, mlq = ml >> 2;
   const int R0 = grp_row0[g];
   const int cl = half * hw + (lane & (hw - 1));             // this lane's column within the group
   const int s = col_slot[g * gw + cl], len = col_slot[(ngrp + g) * gw + cl];
   const float4 *f4 = reinterpret_cast<const float4 *> (fac_t + grp_base[g]) + cl;
#define LDSP_FWD_K0(t) (((t) < nch ? (t) : nch - 1) * CH)
#define LDSP_BWD_K0(t) ((nch - 1 - (t) > 0 ? nch - 1 - (t) : 0) * CH)
   PackChunk<P> F[NCH], Bq[NCH];
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.f_upfront) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
   for (int t = 0; t < NCH; t++)
      if (t < S.b_upfront) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
   // slot (system q, column c) at (q * 16 + c) * STRIDE; one column per step, lane = row, the row's four values in two 16-byte loads
#pragma unroll
   for (int c0 = 0; c0 < hw; c0 += 4) {
      double2 ta[4], tb[4], ua[4], ub[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         const int sc = __builtin_amdgcn_readlane (s, c0 + u), lc = __builtin_amdgcn_readlane (len, c0 + u);
         const double *src = rhs + ((int64_t) R0 + sc + lane) * K;
         ta[u] = (lane < lc) ? *reinterpret_cast<const double2 *> (src) : make_double2 (0.0, 0.0);
         tb[u] = (lane < lc) ? *reinterpret_cast<const double2 *> (src + 2) : make_double2 (0.0, 0.0);
         if (NCH > 4) {
            ua[u] = (lane + NKP_WAVE < lc) ? *reinterpret_cast<const double2 *> (src + (int64_t) NKP_WAVE * K) : make_double2 (0.0, 0.0);
            ub[u] = (lane + NKP_WAVE < lc) ? *reinterpret_cast<const double2 *> (src + (int64_t) NKP_WAVE * K + 2) : make_double2 (0.0, 0.0);
         }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         lds[(0 * hw + c0 + u) * STRIDE + lane] = ta[u].x;
         lds[(1 * hw + c0 + u) * STRIDE + lane] = ta[u].y;
         lds[(2 * hw + c0 + u) * STRIDE + lane] = tb[u].x;
         lds[(3 * hw + c0 + u) * STRIDE + lane] = tb[u].y;
         if (NCH > 4 && lane < STRIDE - 1 - NKP_WAVE) {
            lds[(0 * hw + c0 + u) * STRIDE + lane + NKP_WAVE] = ua[u].x;
            lds[(1 * hw + c0 + u) * STRIDE + lane + NKP_WAVE] = ua[u].y;
            lds[(2 * hw + c0 + u) * STRIDE + lane + NKP_WAVE] = ub[u].x;
            lds[(3 * hw + c0 + u) * STRIDE + lane + NKP_WAVE] = ub[u].y;
         }
      }
   }
   __syncthreads ();
   {
      double *col = lds + lane * STRIDE;                    // slot lane = (system, column)
      double w[P];
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
         if (c < nch) ldsp_step_fwd<P> (F[c], col, c * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.f_after[c - 1] : S.f_upfront) && t < S.f_after[c]) ldsp_load_fwd<P> (F[t], f4, mlq, LDSP_FWD_K0 (t), gw);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (c ? S.b_after_f[c - 1] : S.b_upfront) && t < S.b_after_f[c]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
#pragma unroll
      for (int q = 0; q < P; q++) w[q] = 0.0;
#pragma unroll
      for (int j = 0; j < NCH; j++) {
         if (j < nch) ldsp_step_bwd<P> (Bq[j], col, (nch - 1 - j) * CH, w);
#pragma unroll
         for (int t = 0; t < NCH; t++)
            if (t >= (j ? S.b_after_b[j - 1] : S.b_after_f[NCH - 1]) && t < S.b_after_b[j]) ldsp_load_bwd<P> (Bq[t], f4, mlq, LDSP_BWD_K0 (t), gw);
      }
   }
#undef LDSP_FWD_K0
#undef LDSP_BWD_K0
   __syncthreads ();
#pragma unroll
   for (int c0 = 0; c0 < hw; c0 += 4) {
      double2 ta[4], tb[4], ua[4], ub[4];
      int sc[4], lc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
         sc[u] = __builtin_amdgcn_readlane (s, c0 + u);
         lc[u] = __builtin_amdgcn_readlane (len, c0 + u);
         const double *src = z + ((int64_t) R0 + sc[u] + lane) * K;
         ta[u] = (accumulate && lane < lc[u]) ? *reinterpret_cast<const double2 *> (src) : make_double2 (0.0, 0.0);
         tb[u] = (accumulate && lane < lc[u]) ? *reinterpret_cast<const double2 *> (src + 2) : make_double2 (0.0, 0.0);
         if (NCH > 4) {
            ua[u] = (accumulate && lane + NKP_WAVE < lc[u]) ? *reinterpret_cast<const double2 *> (src + (int64_t) NKP_WAVE * K) : make_double2 (0.0, 0.0);
            ub[u] = (accumulate && lane + NKP_WAVE < lc[u]) ? *reinterpret_cast<const double2 *> (src + (int64_t) NKP_WAVE * K + 2) : make_double2 (0.0, 0.0);
         }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
         double *dst = z + ((int64_t) R0 + sc[u] + lane) * K;
         if (lane < lc[u]) {
            *reinterpret_cast<double2 *> (dst) = make_double2 (ta[u].x + lds[(0 * hw + c0 + u) * STRIDE + lane], ta[u].y + lds[(1 * hw + c0 + u) * STRIDE + lane]);
            *reinterpret_cast<double2 *> (dst + 2) = make_double2 (tb[u].x + lds[(2 * hw + c0 + u) * STRIDE + lane], tb[u].y + lds[(3 * hw + c0 + u) * STRIDE + lane]);
         }
         if (NCH > 4 && lane + NKP_WAVE < lc[u]) {
            double *d2 = dst + (int64_t) NKP_WAVE * K;
            *reinterpret_cast<double2 *> (d2) = make_double2 (ua[u].x + lds[(0 * hw + c0 + u) * STRIDE + lane + NKP_WAVE], ua[u].y + lds[(1 * hw + c0 + u) * STRIDE + lane + NKP_WAVE]);
            *reinterpret_cast<double2 *> (d2 + 2) = make_double2 (ub[u].x + lds[(2 * hw + c0 + u) * STRIDE + lane + NKP_WAVE], ub[u].y + lds[(3 * hw + c0 + u) * STRIDE + lane + NKP_WAVE]);
         }
      }
   }
}

// groups [g0, g1) of a level whose layout is the packed one (B.ldsres == 2), K-interleaved right-hand sides: K / 2 launches of the
// two-system kernel.  Returns non-zero (nothing launched) for any other layout: the caller then uses the wave-per-column batch kernel.
int launch_colblock_apply_lanes_batch (int K, const ColBlocksDev &B, int g0, int g1, const double *r, double *z, int accumulate, hipStream_t st)
{
   if (B.ldsres != 2 || !B.fac_tf) return 1;
   if (g1 <= g0) return 0;
   const int nch = B.max_len <= 64 ? 4 : 5;
   const size_t lds = (size_t) 64 * (size_t) (nch * NKP_LDSRES_CH + 1) * sizeof (double);
   if (K % 4 == 0) {
      // one launch per four systems, two waves (16 columns each) per group
      for (int k0 = 0; k0 < K; k0 += 4) {
#define LDSP4_LAUNCH(PP) do { if (nch == 4) hipLaunchKernelGGL ((colblock_apply_ldspack4_kernel<PP, 4>), dim3 (2 * (g1 - g0)), dim3 (NKP_WAVE), lds, st, B.grp_maxlen, B.grp_base, g0, \
                                                                 B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp, K, k0);                                         \
                              else hipLaunchKernelGGL ((colblock_apply_ldspack4_kernel<PP, 5>), dim3 (2 * (g1 - g0)), dim3 (NKP_WAVE), lds, st, B.grp_maxlen, B.grp_base, g0,  \
                                                       B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp, K, k0); } while (0)
         if (B.P == 1) LDSP4_LAUNCH (1);
         else if (B.P == 2) LDSP4_LAUNCH (2);
         else LDSP4_LAUNCH (4);
#undef LDSP4_LAUNCH
      }
      return 0;
   }
   for (int k0 = 0; k0 < K; k0 += 2) {
#define LDSP2_LAUNCH(PP) do { if (nch == 4) hipLaunchKernelGGL ((colblock_apply_ldspack2_kernel<PP, 4>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_maxlen, B.grp_base, g0, \
                                                                 B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp, K, k0);                                         \
                              else hipLaunchKernelGGL ((colblock_apply_ldspack2_kernel<PP, 5>), dim3 (g1 - g0), dim3 (NKP_WAVE), lds, st, B.grp_maxlen, B.grp_base, g0,       \
                                                       B.fac_tf, r, z, accumulate, B.grp_row0, B.col_slot, B.ngrp, K, k0); } while (0)
      if (B.P == 1) LDSP2_LAUNCH (1);
      else if (B.P == 2) LDSP2_LAUNCH (2);
      else LDSP2_LAUNCH (4);
#undef LDSP2_LAUNCH
   }
   return 0;
}
