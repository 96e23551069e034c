// Device-side construction of the multilevel hierarchy (see mlsetup.hip): the row- and entry-level passes of the setup
// as HIP kernels.  Column-level decisions (a few 10^4 .. 10^5 items per level) stay on the host, in multilevel.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace mls {

// device buffer owned by a scope
template <class T>
struct DBuf {
   T *p = nullptr;
   size_t cnt = 0;
   DBuf () = default;
   DBuf (const DBuf &) = delete;
   DBuf &operator= (const DBuf &) = delete;
   ~DBuf () { reset (); }
   void reset () { if (p) (void) hipFree (p); p = nullptr; cnt = 0; }
   hipError_t alloc (size_t c)
   {
      reset ();
      void *q = nullptr;
      const hipError_t e = hipMalloc (&q, (c ? c : 1) * sizeof (T));
      if (e == hipSuccess) { p = (T *) q; cnt = c; }
      return e;
   }
   T *release () { T *q = p; p = nullptr; cnt = 0; return q; }
};

// CSR in the natural (water-column) row order of a level; setup only
struct DevCsr {
   int64_t n = 0, nnz = 0;
   int *rowptr = nullptr, *colind = nullptr;
   double *val = nullptr;
   void free_all ()
   {
      if (rowptr) (void) hipFree (rowptr);
      if (colind) (void) hipFree (colind);
      if (val) (void) hipFree (val);
      rowptr = colind = nullptr;
      val = nullptr;
      n = nnz = 0;
   }
};

// all functions return 0 or a hipError_t cast to int (1000 + code for logic errors); work is enqueued on st, functions that
// report a host value synchronise st

// out[0..n] = exclusive prefix sums of in[0..n-1] (out[n] = total); in == out allowed (out needs n + 1 entries)
int scan_exclusive (const int *d_in, int *d_out, int64_t n, hipStream_t st, int64_t *total_host /* may be NULL */);

// col_of[r] = column block of row r
int rows_to_cols (const int *d_blk_start, int ncol, int *d_col_of, hipStream_t st);

// low-order twin of A (multilevel.hip: build_low_order), natural order
int twin (int64_t n, const int *d_rowptr, const int *d_colind, const double *d_val, const int *d_col_of, DevCsr &L, hipStream_t st);

// ---- connectivity-aware coarse cells of one coarsening step (multilevel.hip: split_aggregate)
struct AggregateIn {
   int64_t n = 0;
   int ncol = 0;
   const int *rowptr = nullptr, *colind = nullptr;      // device: level operator, natural order
   const double *val = nullptr;
   const int *blk_start = nullptr, *col_of = nullptr, *ktop = nullptr;   // device
   const int *h_blk_start = nullptr, *h_ktop = nullptr, *h_group = nullptr, *h_col_t = nullptr;   // host, per column
   int pocket = 4;
   double tau = 0.01;
};
struct AggregateOut {
   int *cmap = nullptr;                      // device, fine row -> coarse row (caller frees)
   std::vector<int> blk_start, ktop, group;  // coarse columns (host): row offsets, first depth, group the column sits at
   int absorbed = 0, stubs = 0;
};
int aggregate (const AggregateIn &in, AggregateOut &out, hipStream_t st);

// coarse row -> its fine rows, ascending inside a segment (stable counting sort of the rows by map[r])
int inverse_map (const int *d_map, int64_t n, int64_t nc, int **d_rptr, int **d_ridx, hipStream_t st);

// C = P^T L P for the piecewise-constant P given by cmap; every coarse entry is the sum of its fine entries in ascending
// (fine row, stored position) order; exact zeros are not stored except on the diagonal
int galerkin (const DevCsr &L, const int *d_cmap, int64_t nc, DevCsr &C, hipStream_t st);

// rows reordered (new row i = old row perm[i]), columns relabelled through inv and sorted ascending
int permute_operator (const DevCsr &L, const int *d_perm, const int *d_inv, int **d_prow, int **d_pcol, double **d_pval, int64_t pad, hipStream_t st);

int to_float (const double *d_src, float *d_dst, int64_t cnt, hipStream_t st);

// perm / inv of the colour-major row order from the new start row of every column
int colour_major_maps (const int *d_blk_start, const int *d_col_of, const int *d_newstart /* per column */, int64_t n, int *d_perm, int *d_inv, hipStream_t st);

// cmap_p[i] = inv_coarse[cmap[perm[i]]]
int permuted_cmap (const int *d_cmap, const int *d_perm, const int *d_inv_coarse, int64_t n, int *d_out, hipStream_t st);

// CSR-stream row blocks (spmv.hip) of rows [r0, r1): boundaries (absolute rows, first = r0, last = r1) appended to *d_out
// (device array the function allocates), count in *nblocks
int row_blocks (const int *d_rowptr, int64_t r0, int64_t r1, int **d_out, int *nblocks, hipStream_t st);

}  // namespace mls
