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
#define SPMV_LDS_NNZ NKP_SPMV_LDS_NNZ
#define SPMV_MAX_ROWS NKP_SPMV_MAX_ROWS

template <int MODE, int VAR, class VT>   // MODE 0: y = A x   1: y = b - A x   2: y = |A||x| + |b| ; VAR: load flavour (5-8: ablation build only) ; VT: stored value type
__global__ __launch_bounds__ (SPMV_THREADS)
void csr_spmv_stream_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk, int per_xcd,
                             const int *__restrict__ rowptr, const int *__restrict__ colind,
                             const VT *__restrict__ val, const double *__restrict__ x,
                             double *__restrict__ y, const double *__restrict__ b,
                             const unsigned short *__restrict__ codes, const int *__restrict__ dict, const int *__restrict__ dict_ptr)
{
   __shared__ double prod[SPMV_LDS_NNZ];
   __shared__ int dict_s[256];
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
   // this lane's row bounds for phase 2, requested now so their latency hides under phase 1
   int seg0 = 0, seg1 = 0;
   if (r0 + tid < r1) { seg0 = rowptr[r0 + tid]; seg1 = rowptr[r0 + tid + 1]; }

   if (cnt > SPMV_LDS_NNZ) {
      // a single long row (the partitioner never packs several rows past the LDS budget)
      double acc = 0.0;
      for (int e = e0 + tid; e < e1; e += SPMV_THREADS) {
         double p = (double) val[e] * x[colind[e]];
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

#ifdef NKP_ABLATION
   if (VAR == 8) {
      // TIMING-ONLY: pure stream of the two arrays, no gather, no LDS, no row sums
      double acc = 0.0;
#pragma unroll 8
      for (int k = tid; k < cnt; k += SPMV_THREADS) acc += (double) val[e0 + k] * (double) colind[e0 + k];
      if (acc == 123.456) y[r0] = acc;
      return;
   }
#endif
   // phase 1: coalesced stream of (val, colind), gather x through L2, products staged in LDS
   if (VAR == 4) {
      // 2-byte column codes: 10 instead of 12 bytes per entry off HBM
      const int d0 = dict_ptr[rb0 + lb], nd = dict_ptr[rb0 + lb + 1] - d0;
      if (nd > 0) {
         for (int i = tid; i < nd; i += SPMV_THREADS) dict_s[i] = dict[d0 + i];
         __syncthreads ();
#pragma unroll 8
         for (int k = tid; k < cnt; k += SPMV_THREADS) {
            const unsigned int code = codes[e0 + k];
            const int c = r0 + (int) (code & 255u) + dict_s[code >> 8];
            const double q = (double) val[e0 + k] * x[c];
            prod[k] = (MODE == 2) ? fabs (q) : q;
         }
      } else {
#pragma unroll 8
         for (int k = tid; k < cnt; k += SPMV_THREADS) {
            const double q = (double) val[e0 + k] * x[colind[e0 + k]];
            prod[k] = (MODE == 2) ? fabs (q) : q;
         }
      }
   } else if (VAR & 1) {
      // 16-byte / 8-byte pairs: the block's entry range is widened to even boundaries, out-of-range halves masked
      typedef double dbl2_t __attribute__ ((ext_vector_type (2)));
      typedef int int2_t __attribute__ ((ext_vector_type (2)));
      const int a0 = e0 & ~1;
      const int npair = ((e1 + 1) >> 1) - (a0 >> 1);
      const dbl2_t *val2 = reinterpret_cast<const dbl2_t *> (reinterpret_cast<const double *> (val) + a0);
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
#ifdef NKP_ABLATION
   } else if (VAR >= 5) {
      // TIMING-ONLY ablations (wrong results by design): 5 = no x gather, 6 = no row sums, 7 = no value stream
#pragma unroll 8
      for (int k = tid; k < cnt; k += SPMV_THREADS) {
         const double v = (VAR == 7) ? 1.0 : (double) val[e0 + k];
         const int c = colind[e0 + k];
         const double q = v * ((VAR == 5) ? (double) c : x[c]);
         prod[k] = q;
      }
#endif
   } else {
#pragma unroll 8
      for (int k = tid; k < cnt; k += SPMV_THREADS) {
         const double v = (double) ((VAR & 2) ? __builtin_nontemporal_load (val + e0 + k) : val[e0 + k]);
         const int c = (VAR & 2) ? __builtin_nontemporal_load (colind + e0 + k) : colind[e0 + k];
         const double q = v * x[c];
         prod[k] = (MODE == 2) ? fabs (q) : q;
      }
   }
   __syncthreads ();
#ifdef NKP_ABLATION
   if (VAR == 6) {
      if (tid == 0) y[r0] = prod[0] + prod[cnt - 1];
      return;
   }
#endif

   // phase 2: one lane per row sums its LDS segment (fixed order => deterministic)
   const int r = r0 + tid;
   if (r < r1) {
      const int s0 = seg0 - e0, s1 = seg1 - e0;
      double acc = 0.0;
#pragma unroll 4
      for (int k = s0; k < s1; k++) acc += prod[k];
      if (MODE == 1) acc = b[r] - acc;
      if (MODE == 2) acc += fabs (b[r]);
      y[r] = acc;
   }
}

// ---------------------------------------------------------------- software-pipelined variant
// Ablations on the 1 degree matrix (rocprof, this file's VAR 5-8): the bare (val, colind) stream runs at
// 6.1 TB/s, but a workgroup that streams, THEN gathers x, THEN sums rows exposes three latencies in
// series and the whole kernel lands at 4.2 TB/s.  Here a persistent workgroup walks a contiguous run of
// row blocks and requests block i+1's stream (24 registers per lane) before it consumes block i's
// gathers, so the stream latency hides under the gather + LDS + row-sum phases of the previous block.
// Same products, same per-row summation order => bit-identical results.
#define SPMV_SLOTS (SPMV_LDS_NNZ / SPMV_THREADS)

// CODED: the column stream is read as 2-byte codes (build_spmv_codes_host: row-in-block in the low byte, index into the
// row block's dictionary of (column - row) offsets in the high byte) -- 10 instead of 12 bytes per entry for the f64
// operator, 6 instead of 8 for the f32 level operators.  The next block's dictionary is fetched into the other half of
// a double-buffered LDS table while the current block is consumed.  Blocks the coder gave up on (more than 256
// distinct offsets, long rows) have an empty dictionary and fall back to colind.
template <int MODE, class VT, bool CODED>
__global__ __launch_bounds__ (SPMV_THREADS)
void csr_spmv_pipe_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk,
                           const int *__restrict__ rowptr, const int *__restrict__ colind,
                           const VT *__restrict__ val, const double *__restrict__ x,
                           double *__restrict__ y, const double *__restrict__ b,
                           const unsigned short *__restrict__ codes, const int *__restrict__ dict, const int *__restrict__ dict_ptr)
{
   __shared__ double prod[SPMV_LDS_NNZ];
   __shared__ double wsum[SPMV_THREADS / NKP_WAVE];
   __shared__ int dict_s[CODED ? 2 : 1][CODED ? 256 : 1];
   const int tid = threadIdx.x;
   const int *rowblk = rowblk_all + rb0;
   // XCD-aware contiguous run of row blocks for this workgroup
   const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
   const int per_xcd = (nrowblk + 7) / 8;
   const int xb0 = min (xcd * per_xcd, nrowblk), xb1 = min (xb0 + per_xcd, nrowblk);
   const int chunk = (xb1 - xb0 + wg_per_xcd - 1) / wg_per_xcd;
   int lb = xb0 + idx * chunk;
   const int lb_end = min (lb + chunk, xb1);
   if (lb >= lb_end) return;

   int r0 = rowblk[lb], r1 = rowblk[lb + 1];
   int e0 = rowptr[r0], e1 = rowptr[r1];
   VT v[SPMV_SLOTS];
   int c[SPMV_SLOTS];           // column, or its code while the block waits to be consumed
   int seg0 = 0, seg1 = 0;
   int nd = 0;                  // dictionary length of the current block (0: columns are plain)
   int cur = 0;                 // which half of dict_s holds it

   // synchronous load of block lb into the registers (first block, and after a long-row block)
#define SPMV_LOAD_CURRENT()                                                                        \
   do {                                                                                            \
      int d0_ = 0;                                                                                 \
      nd = 0;                                                                                      \
      if (CODED) { d0_ = dict_ptr[rb0 + lb]; nd = dict_ptr[rb0 + lb + 1] - d0_; }                  \
      _Pragma ("unroll")                                                                           \
      for (int u = 0; u < SPMV_SLOTS; u++) {                                                       \
         const int e = e0 + tid + u * SPMV_THREADS;                                                \
         const bool ok = e < e1 && e - e0 < SPMV_LDS_NNZ;                                          \
         v[u] = ok ? val[e] : (VT) 0;                                                              \
         c[u] = ok ? ((CODED && nd > 0) ? (int) codes[e] : colind[e]) : 0;                         \
      }                                                                                            \
      seg0 = seg1 = 0;                                                                             \
      if (r0 + tid < r1) { seg0 = rowptr[r0 + tid]; seg1 = rowptr[r0 + tid + 1]; }                \
      if (CODED) {                                                                                 \
         if (tid < nd) dict_s[cur][tid] = dict[d0_ + tid];                                         \
         __syncthreads ();                                                                         \
      }                                                                                            \
   } while (0)

   SPMV_LOAD_CURRENT ();

   for (;;) {
      const int cnt = e1 - e0;
      const bool have_next = lb + 1 < lb_end;
      // descriptors of the next block (it starts where this one ends)
      int nr1 = r1, ne1 = e1;
      if (have_next) { nr1 = rowblk[lb + 2]; ne1 = rowptr[nr1]; }

      if (cnt > SPMV_LDS_NNZ) {
         // a single long row: strided accumulate + block reduction (tree order)
         double acc = 0.0;
         for (int e = e0 + tid; e < e1; e += SPMV_THREADS) {
            const double p = (double) val[e] * x[colind[e]];
            acc += (MODE == 2) ? fabs (p) : p;
         }
         for (int off = NKP_WAVE / 2; off > 0; off >>= 1) acc += __shfl_down (acc, off);
         if ((tid & (NKP_WAVE - 1)) == 0) wsum[tid / NKP_WAVE] = acc;
         __syncthreads ();
         if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < SPMV_THREADS / NKP_WAVE; w++) t += wsum[w];
            if (MODE == 1) t = b[r0] - t;
            if (MODE == 2) t += fabs (b[r0]);
            y[r0] = t;
         }
         __syncthreads ();
      } else {
         // decode, gather this block, request the NEXT block's streams (and dictionary), then consume the gathers
         if (CODED && nd > 0) {
#pragma unroll
            for (int u = 0; u < SPMV_SLOTS; u++) c[u] = r0 + (c[u] & 255) + dict_s[cur][(c[u] >> 8) & 255];
         }
         double xg[SPMV_SLOTS];
#pragma unroll
         for (int u = 0; u < SPMV_SLOTS; u++) xg[u] = (tid + u * SPMV_THREADS < cnt) ? x[c[u]] : 0.0;
         VT nv[SPMV_SLOTS];
         int nc[SPMV_SLOTS];
         int nseg0 = 0, nseg1 = 0;
         int nnd = 0, nd0 = 0;
         if (CODED && have_next) { nd0 = dict_ptr[rb0 + lb + 1]; nnd = dict_ptr[rb0 + lb + 2] - nd0; }
#pragma unroll
         for (int u = 0; u < SPMV_SLOTS; u++) {
            const int e = e1 + tid + u * SPMV_THREADS;
            const bool ok = have_next && e < ne1 && e - e1 < SPMV_LDS_NNZ;
            nv[u] = ok ? val[e] : (VT) 0;
            nc[u] = ok ? ((CODED && nnd > 0) ? (int) codes[e] : colind[e]) : 0;
         }
         if (have_next && r1 + tid < nr1) { nseg0 = rowptr[r1 + tid]; nseg1 = rowptr[r1 + tid + 1]; }
         // the other half of dict_s was last read while block lb - 1 was decoded, two barriers ago
         if (CODED && tid < nnd) dict_s[cur ^ 1][tid] = dict[nd0 + tid];
#pragma unroll
         for (int u = 0; u < SPMV_SLOTS; u++) {
            const int k = tid + u * SPMV_THREADS;
            if (k < cnt) {
               const double q = (double) v[u] * xg[u];
               prod[k] = (MODE == 2) ? fabs (q) : q;
            }
         }
         __syncthreads ();
         const int r = r0 + tid;
         if (r < r1) {
            const int s0 = seg0 - e0, s1 = seg1 - e0;
            double acc = 0.0;
#pragma unroll 4
            for (int k = s0; k < s1; k++) acc += prod[k];
            if (MODE == 1) acc = b[r] - acc;
            if (MODE == 2) acc += fabs (b[r]);
            y[r] = acc;
         }
         __syncthreads ();
#pragma unroll
         for (int u = 0; u < SPMV_SLOTS; u++) { v[u] = nv[u]; c[u] = nc[u]; }
         seg0 = nseg0;
         seg1 = nseg1;
         nd = nnd;
         cur ^= 1;
         if (!have_next) break;
         lb++;
         r0 = r1; r1 = nr1; e0 = e1; e1 = ne1;
         continue;
      }
      // after a long-row block the prefetched registers are stale: reload for the next block
      if (!have_next) break;
      lb++;
      r0 = r1; r1 = nr1; e0 = e1; e1 = ne1;
      SPMV_LOAD_CURRENT ();
   }
#undef SPMV_LOAD_CURRENT
}

// ---------------------------------------------------------------- rows variant (spmv_variant 9)
// Found on the batched kernel (batch.hip) and brought back here: stage the block's (value, column) stream in LDS instead of
// the products -- 8 bytes per entry for an f32 operator -- and let every row's lane walk its own segment, gathering x itself
// with GU loads in flight and summing in registers.  No second pass over LDS with 8-byte products, half the LDS per
// workgroup, and the gathers of a wave go to few neighbouring lines (lanes = consecutive rows of a column: their stencil
// neighbours are consecutive too).  Same products, same stored order => same bits.
template <int MODE, class VT>
__global__ __launch_bounds__ (SPMV_THREADS)
void csr_spmv_rows_kernel (const int *__restrict__ rowblk_all, int rb0, int nrowblk, int per_xcd,
                           const int *__restrict__ rowptr, const int *__restrict__ colind,
                           const VT *__restrict__ val, const double *__restrict__ x,
                           double *__restrict__ y, const double *__restrict__ b)
{
   __shared__ VT sv[SPMV_LDS_NNZ];
   __shared__ int sc[SPMV_LDS_NNZ];
   __shared__ double wsum[SPMV_THREADS / NKP_WAVE];
   constexpr int GU = 8;
   const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
   const int lb = xcd * per_xcd + idx;
   if (idx >= per_xcd || lb >= nrowblk) return;
   const int tid = threadIdx.x;
   const int *rowblk = rowblk_all + rb0;
   const int r0 = rowblk[lb], r1 = rowblk[lb + 1];
   const int e0 = rowptr[r0], e1 = rowptr[r1];
   const int cnt = e1 - e0;
   if (cnt > SPMV_LDS_NNZ) {
      double acc = 0.0;
      for (int e = e0 + tid; e < e1; e += SPMV_THREADS) {
         double p = (double) val[e] * x[colind[e]];
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
   const int r = r0 + tid;
   int s0 = 0, s1 = 0;
   if (r < r1) { s0 = rowptr[r] - e0; s1 = rowptr[r + 1] - e0; }
   {
      VT v[SPMV_LDS_NNZ / SPMV_THREADS];
      int c[SPMV_LDS_NNZ / SPMV_THREADS];
#pragma unroll
      for (int u = 0; u < SPMV_LDS_NNZ / SPMV_THREADS; u++) {
         const int k = tid + u * SPMV_THREADS;
         v[u] = k < cnt ? val[e0 + k] : (VT) 0;
         c[u] = k < cnt ? colind[e0 + k] : 0;
      }
#pragma unroll
      for (int u = 0; u < SPMV_LDS_NNZ / SPMV_THREADS; u++) {
         const int k = tid + u * SPMV_THREADS;
         if (k < cnt) { sv[k] = v[u]; sc[k] = c[u]; }
      }
   }
   const double bv = (MODE != 0 && r < r1) ? b[r] : 0.0;
   __syncthreads ();
   if (r >= r1) return;
   double acc = 0.0;
   for (int k = s0; k < s1; k += GU) {
      double vv[GU], xg[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) {
         const int kk = k + u < s1 ? k + u : s1 - 1;
         vv[u] = (double) sv[kk];
         xg[u] = x[sc[kk]];
      }
#pragma unroll
      for (int u = 0; u < GU; u++)
         if (k + u < s1) {
            const double q = vv[u] * xg[u];
            acc += (MODE == 2) ? fabs (q) : q;
         }
   }
   if (MODE == 1) acc = bv - acc;
   if (MODE == 2) acc += fabs (bv);
   y[r] = acc;
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

template <int MODE>
static void launch_range (const CsrDev &A, int rb0, int cnt, const double *x, double *y, const double *b, hipStream_t st)
{
   if (cnt <= 0) return;
   const int per_xcd = (cnt + 7) / 8;
   const nkp_tuning &T = A.tune ? *A.tune : nkp_builtin_tuning ();
   if (T.spmv_variant == 4 && cnt >= T.spmv_pipe_min) {
      // how many row blocks one workgroup walks (spmv_run) and how many workgroups per CU at most (spmv_wgs).
      // Same-process A/B at 1 degree, first with runs of >= 3: 6 per CU 199 us, 24 per CU 184 us, 48 per CU 177 us; then
      // runs of 1 with 160-256 per CU: SpMV -1..2 %, V-cycle 2.71 -> 2.63 ms, Arnoldi step (j = 100) 7.02 -> 6.86 ms on
      // two repeats -- enough workgroups in flight hide the stream latency as well as the in-workgroup prefetch does
      const int per_cu = T.spmv_wgs > 0 ? T.spmv_wgs : 256, run_len = T.spmv_run > 0 ? T.spmv_run : 1;
      int wgs = cnt / run_len;
      if (wgs > 256 * per_cu) wgs = 256 * per_cu;
      wgs &= ~7;
      if (wgs < 8) wgs = 8;
#define SPMV_PIPE(VT_, CODED_, VAL_) hipLaunchKernelGGL ((csr_spmv_pipe_kernel<MODE, VT_, CODED_>), dim3 (wgs), dim3 (SPMV_THREADS), 0, st, \
                                                         A.rowblk, rb0, cnt, A.rowptr, A.colind, VAL_, x, y, b, A.codes, A.dict, A.dict_ptr)
      if (A.valf) { if (A.codes) SPMV_PIPE (float, true, A.valf); else SPMV_PIPE (float, false, A.valf); }
      else { if (A.codes) SPMV_PIPE (double, true, A.val); else SPMV_PIPE (double, false, A.val); }
#undef SPMV_PIPE
      return;
   }
   if (T.spmv_variant == 9) {
      if (A.valf) hipLaunchKernelGGL ((csr_spmv_rows_kernel<MODE, float>), dim3 (per_xcd * 8), dim3 (SPMV_THREADS), 0, st, A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, A.valf, x, y, b);
      else hipLaunchKernelGGL ((csr_spmv_rows_kernel<MODE, double>), dim3 (per_xcd * 8), dim3 (SPMV_THREADS), 0, st, A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, A.val, x, y, b);
      return;
   }
   if (A.valf) {
      hipLaunchKernelGGL ((csr_spmv_stream_kernel<MODE, 0, float>), dim3 (per_xcd * 8), dim3 (SPMV_THREADS), 0, st,
                          A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, A.valf, x, y, b, A.codes, A.dict, A.dict_ptr);
      return;
   }
#define SPMV_GO(VV) hipLaunchKernelGGL ((csr_spmv_stream_kernel<MODE, VV, double>), dim3 (per_xcd * 8), dim3 (SPMV_THREADS), 0, st, \
                                         A.rowblk, rb0, cnt, per_xcd, A.rowptr, A.colind, A.val, x, y, b, A.codes, A.dict, A.dict_ptr)
   int var = T.spmv_variant;
   if (var == 4 && !A.codes) var = 0;
   switch (var) {
   case 1: SPMV_GO (1); break;
   case 2: SPMV_GO (2); break;
   case 3: SPMV_GO (3); break;
   case 4: SPMV_GO (4); break;
#ifdef NKP_ABLATION
   // timing-only bodies that produce WRONG results by design: never in libnkp_hip.so, only in the `make ablation` library
   case 5: SPMV_GO (5); break;
   case 6: SPMV_GO (6); break;
   case 7: SPMV_GO (7); break;
   case 8: SPMV_GO (8); break;
#endif
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

void launch_csr_spmv_range (const CsrDev &A, int rb0, int rb1, const double *x, double *y, const double *b, int mode, hipStream_t st)
{
   if (mode == 0) launch_range<0> (A, rb0, rb1 - rb0, x, y, nullptr, st);
   else if (mode == 1) launch_range<1> (A, rb0, rb1 - rb0, x, y, b, st);
   else launch_range<2> (A, rb0, rb1 - rb0, x, y, b, st);
}

// ---------------------------------------------------------------- 2-byte column codes (host build)
#include <algorithm>
#include <vector>

double build_spmv_codes_host (int64_t n, const int *rowptr, const int *colind, const int *rowblk, int nrowblk,
                              unsigned short **codes_out, int **dict_out, int *ndict_out, int **dict_ptr_out)
{
   const int64_t nnz = rowptr[n];
   unsigned short *codes = (unsigned short *) malloc ((size_t) (nnz + 2) * sizeof (unsigned short));
   int *dict_ptr = (int *) malloc ((size_t) (nrowblk + 1) * sizeof (int));
   std::vector<int> dict;
   std::vector<int> deltas;
   int64_t coded = 0;
   dict_ptr[0] = 0;
   for (int b = 0; b < nrowblk; b++) {
      const int r0 = rowblk[b], r1 = rowblk[b + 1];
      const int e0 = rowptr[r0], e1 = rowptr[r1];
      bool ok = (r1 - r0) <= 256 && (e1 - e0) <= SPMV_LDS_NNZ;
      if (ok) {
         deltas.clear ();
         for (int r = r0; r < r1; r++)
            for (int e = rowptr[r]; e < rowptr[r + 1]; e++) deltas.push_back (colind[e] - r);
         std::sort (deltas.begin (), deltas.end ());
         deltas.erase (std::unique (deltas.begin (), deltas.end ()), deltas.end ());
         ok = deltas.size () <= 256 && !deltas.empty ();
      }
      if (ok) {
         for (int r = r0; r < r1; r++)
            for (int e = rowptr[r]; e < rowptr[r + 1]; e++) {
               const int id = (int) (std::lower_bound (deltas.begin (), deltas.end (), colind[e] - r) - deltas.begin ());
               codes[e] = (unsigned short) ((r - r0) | (id << 8));
            }
         dict.insert (dict.end (), deltas.begin (), deltas.end ());
         coded += e1 - e0;
      } else
         for (int e = e0; e < e1; e++) codes[e] = 0;
      dict_ptr[b + 1] = (int) dict.size ();
   }
   int *d = (int *) malloc ((dict.size () + 1) * sizeof (int));
   std::copy (dict.begin (), dict.end (), d);
   *codes_out = codes;
   *dict_out = d;
   *ndict_out = (int) dict.size ();
   *dict_ptr_out = dict_ptr;
   return nnz ? (double) coded / (double) nnz : 1.0;
}

int attach_spmv_codes (CsrDev &A, const int *h_rowptr, const int *h_colind, const int *h_rowblk, size_t *device_bytes)
{
   // off by default (nkp_tuning.spmv_compress / NKP_SPMV_COMPRESS=1 turns it on): measured at 1 degree the 2-byte codes do not
   // shorten the kernels -- pipelined f64 SpMV 0.236 ms coded against 0.203 ms plain on the same box, V-cycle 2.93 against
   // 2.69 ms: the dictionary lookup sits in front of the x gather and lengthens the dependent chain by more than the 2 bytes
   // per entry save -- and building them costs ~2.7 s of setup
   if (!(A.tune ? A.tune->spmv_compress : 0)) return 0;
   if (A.nnz == 0 || A.nrowblk == 0) return 0;
   unsigned short *codes = nullptr;
   int *dict = nullptr, *dict_ptr = nullptr, nd = 0;
   build_spmv_codes_host (A.n, h_rowptr, h_colind, h_rowblk, A.nrowblk, &codes, &dict, &nd, &dict_ptr);
   int rc = 0;
   const size_t cb = (size_t) (A.nnz + 2) * sizeof (unsigned short), db = (size_t) (nd + 1) * sizeof (int), pb = (size_t) (A.nrowblk + 1) * sizeof (int);
   if (hipMalloc ((void **) &A.codes, cb) != hipSuccess || hipMalloc ((void **) &A.dict, db) != hipSuccess || hipMalloc ((void **) &A.dict_ptr, pb) != hipSuccess) rc = 1;
   if (!rc && (hipMemcpy (A.codes, codes, cb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy (A.dict, dict, db, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy (A.dict_ptr, dict_ptr, pb, hipMemcpyHostToDevice) != hipSuccess)) rc = 1;
   if (!rc) *device_bytes += cb + db + pb;
   free (codes);
   free (dict);
   free (dict_ptr);
   return rc;
}
