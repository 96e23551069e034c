// CSR SpMV for gfx950: "CSR-stream" -- a workgroup owns a run of consecutive rows whose
// entries fit an LDS staging buffer.  Phase 1 streams the value / column-index arrays with
// fully coalesced loads (lane e reads entry e), gathers x through L2 and parks the products
// in LDS; phase 2 sums each row's contiguous LDS segment.  Rows of an ocean Jacobian hold
// 5-21 entries (reference src/matrix.c:800-961), far too short for a wave per row.
//
// Stands in for pdgsmv_AXglobal (reference src/SuperLU_brief_tree.txt:21-22).
// Roofline: HBM-bound; algorithmic bytes = 12*nnz + 4*(n+1) + 16*n.
//
// XCD-aware launch: consecutive row blocks touch neighbouring x (j+-1 latitude rows are
// ~imt*km rows apart), so each of the 8 XCDs gets one contiguous eighth of the row blocks
// and its private L2 sees the reuse (blocks are dealt round-robin over XCDs, so
// blockIdx % 8 is the group label).
#include "nkp_dev.h"

#include <stdlib.h>

#define SPMV_THREADS 256
#define SPMV_LDS_NNZ 2048
#define SPMV_MAX_ROWS 256

template <int MODE, int VAR>   // MODE 0: y = A x   1: y = b - A x   2: y = |A||x| + |b| ; VAR: load flavour
__global__ __launch_bounds__ (SPMV_THREADS)
void csr_spmv_stream_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk, int per_xcd,
                             const int *__restrict__ rowptr, const int *__restrict__ colind,
                             const double *__restrict__ val, const double *__restrict__ x,
                             double *__restrict__ y, const double *__restrict__ b)
{
   __shared__ double prod[SPMV_LDS_NNZ];
   __shared__ double wsum[SPMV_THREADS / NKP_WAVE];

   const int xcd = blockIdx.x & 7;
   const int idx = blockIdx.x >> 3;
   const int lb = xcd * per_xcd + idx;
   if (idx >= per_xcd || lb >= nrowblk) return;

   const int tid = threadIdx.x;
   const int *rowblk = rowblk_all + rb0;
   const int r0 = rowblk[lb], r1 = rowblk[lb + 1];
   const int e0 = rowptr[r0], e1 = rowptr[r1];
   const int cnt = e1 - e0;

   if (cnt > SPMV_LDS_NNZ) {
      // a single long row (the partitioner never packs several rows past the LDS budget)
      double acc = 0.0;
      for (int e = e0 + tid; e < e1; e += SPMV_THREADS) {
         double p = val[e] * x[colind[e]];
         acc += (MODE == 2) ? fabs (p) : p;
      }
      for (int off = NKP_WAVE / 2; off > 0; off >>= 1) acc += __shfl_down (acc, off);
      if ((tid & (NKP_WAVE - 1)) == 0) wsum[tid / NKP_WAVE] = acc;
      __syncthreads ();
      if (tid == 0) {
         double s = 0.0;
         for (int w = 0; w < SPMV_THREADS / NKP_WAVE; w++) s += wsum[w];
         if (MODE == 1) s = b[r0] - s;
         if (MODE == 2) s += fabs (b[r0]);
         y[r0] = s;
      }
      return;
   }

   // phase 1: coalesced stream of (val, colind), gather x through L2, products staged in LDS
   if (VAR & 1) {
      // 16-byte / 8-byte pairs: the block's entry range is widened to even boundaries, out-of-range halves masked
      typedef double dbl2_t __attribute__ ((ext_vector_type (2)));
      typedef int int2_t __attribute__ ((ext_vector_type (2)));
      const int a0 = e0 & ~1;
      const int npair = ((e1 + 1) >> 1) - (a0 >> 1);
      const dbl2_t *val2 = reinterpret_cast<const dbl2_t *> (val + a0);
      const int2_t *col2 = reinterpret_cast<const int2_t *> (colind + a0);
#pragma unroll 4
      for (int p = tid; p < npair; p += SPMV_THREADS) {
         const dbl2_t v = (VAR & 2) ? __builtin_nontemporal_load (val2 + p) : val2[p];
         const int2_t c = (VAR & 2) ? __builtin_nontemporal_load (col2 + p) : col2[p];
         const int k = a0 + 2 * p - e0;                      // LDS slot of the pair's first entry (may be -1)
         if (k >= 0) {
            const double q = v.x * x[c.x];
            prod[k] = (MODE == 2) ? fabs (q) : q;
         }
         if (k + 1 < cnt) {
            const double q = v.y * x[c.y];
            prod[k + 1] = (MODE == 2) ? fabs (q) : q;
         }
      }
   } else {
#pragma unroll 8
      for (int k = tid; k < cnt; k += SPMV_THREADS) {
         const double v = (VAR & 2) ? __builtin_nontemporal_load (val + e0 + k) : val[e0 + k];
         const int c = (VAR & 2) ? __builtin_nontemporal_load (colind + e0 + k) : colind[e0 + k];
         const double q = v * x[c];
         prod[k] = (MODE == 2) ? fabs (q) : q;
      }
   }
   __syncthreads ();

   // phase 2: one lane per row sums its LDS segment (fixed order => deterministic)
   const int r = r0 + tid;
   if (r < r1) {
      const int s0 = rowptr[r] - e0, s1 = rowptr[r + 1] - e0;
      double acc = 0.0;
      for (int k = s0; k < s1; k++) acc += prod[k];
      if (MODE == 1) acc = b[r] - acc;
      if (MODE == 2) acc += fabs (b[r]);
      y[r] = acc;
   }
}

void build_rowblocks_host (int64_t n, const int *rowptr, int **rowblk_out, int *nrowblk_out)
{
   // worst case one block per row
   int *rb = (int *) malloc ((size_t) (n + 2) * sizeof (int));
   int nb = 0;
   int64_t r = 0;
   rb[0] = 0;
   while (r < n) {
      int64_t r_end = r + 1;      // always take at least one row (possibly a long one)
      int64_t base = rowptr[r];
      while (r_end < n && (r_end - r) < SPMV_MAX_ROWS && (int64_t) rowptr[r_end + 1] - base <= SPMV_LDS_NNZ)
         r_end++;
      // if the very first row alone exceeds the budget it stays alone (long-row path)
      rb[++nb] = (int) r_end;
      r = r_end;
   }
   *rowblk_out = rb;
   *nrowblk_out = nb;
}

static int spmv_variant ()
{
   static int v = -1;
   if (v < 0) {
      const char *e = getenv ("NKP_SPMV_VARIANT");
      v = e ? atoi (e) & 3 : 0;
   }
   return v;
}

template <int MODE>
static void launch_range (const CsrDev &A, int rb0, int cnt, const double *x, double *y, const double *b, hipStream_t st)
{
   if (cnt <= 0) return;
   const int per_xcd = (cnt + 7) / 8;
#define SPMV_GO(VV) hipLaunchKernelGGL ((csr_spmv_stream_kernel<MODE, VV>), dim3 (per_xcd * 8), dim3 (SPMV_THREADS), 0, st, \
                                         A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, A.val, x, y, b)
   switch (spmv_variant ()) {
   case 1: SPMV_GO (1); break;
   case 2: SPMV_GO (2); break;
   case 3: SPMV_GO (3); break;
   default: SPMV_GO (0); break;
   }
#undef SPMV_GO
}

template <int MODE>
static void launch_mode (const CsrDev &A, const double *x, double *y, const double *b, hipStream_t st)
{
   if (A.n == 0) return;
   launch_range<MODE> (A, 0, A.nrowblk, x, y, b, st);
}

void launch_csr_spmv (const CsrDev &A, const double *x, double *y, const double *b, int mode, hipStream_t st)
{
   if (mode == 0) launch_mode<0> (A, x, y, nullptr, st);
   else launch_mode<1> (A, x, y, b, st);
}

void launch_csr_abs_spmv (const CsrDev &A, const double *x, const double *b, double *y, hipStream_t st)
{
   launch_mode<2> (A, x, y, b, st);
}

void launch_csr_residual_range (const CsrDev &A, int rb0, int rb1, const double *x, const double *b, double *y, hipStream_t st)
{
   launch_range<1> (A, rb0, rb1 - rb0, x, y, b, st);
}
