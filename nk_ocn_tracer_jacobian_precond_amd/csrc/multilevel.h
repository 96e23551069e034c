// Multilevel water-column preconditioner: data structures (see multilevel.hip).
#pragma once
#include "nkp_dev.h"

#include <vector>

struct MlLevel {
   int64_t n = 0;
   CsrDev L;                    // level operator, colour-major row order
   ColBlocksDev B;              // factored water-column blocks of L (absent on the coarsest level)
   int color_rb[3] = { 0, 0, 0 };    // SpMV row blocks of colour c: [color_rb[c], color_rb[c+1])
   int color_blk[3] = { 0, 0, 0 };   // column blocks of colour c
   int color_grp[3] = { 0, 0, 0 };   // lane-per-column groups of colour c
   int64_t rows0 = 0;           // rows of colour 0 (they come first)
   int wave_columns = 0;        // few columns: solve them one per wave (colblock_apply_kernel) instead of one per lane
   int wave_fused = 0;          // ... and run every half sweep as one launch (gs_wave_kernel: residual rows + band solve per wave)
   int64_t nc = 0;              // rows of the next coarser level
   int *cmap = nullptr;         // fine row -> coarse row                 (prolongation)
   int *rptr = nullptr, *ridx = nullptr;   // coarse row -> its fine rows (restriction)
   double *x = nullptr, *b = nullptr, *r = nullptr;
   // fused half sweeps ping-pong between x and x2: xs[k] is buffer k, cur[c] the buffer that holds the current values
   // of colour c's rows; the level's x is coherent (one buffer) whenever cur[0] == cur[1]
   double *x2 = nullptr;
   int cur[2] = { 0, 0 };
   // K interleaved right-hand sides (batch.hip): the level's vectors once more, K times as long (allocated on first use)
   double *bx = nullptr, *bx2 = nullptr, *bb = nullptr, *br = nullptr;
   int bcur[2] = { 0, 0 };
   double *bxbuf (int k) { return k ? bx2 : bx; }
   double *bxnow () { return bxbuf (bcur[0]); }
   double *xbuf (int k) { return k ? x2 : x; }
   double *xnow () { return xbuf (cur[0]); }      // valid when coherent
};

struct MlHierarchy {
   std::vector<MlLevel> lev;
   int *perm0 = nullptr;        // level-0 row i holds original row perm0[i]
   double *coarse_inv = nullptr;   // dense inverse of the coarsest operator (row-major)
   float *coarse_invf = nullptr;   // f32 storage mode: the copy the cycle multiplies with (rows padded to coarse_ldf)
   int coarse_ldf = 0;
   int nu = 1;                  // Gauss-Seidel sweeps before and after the coarse correction
   int nu_coarse = 1;           // ... on levels >= coarse_from
   int coarse_from = 2;
   int f32 = 1;                 // store level operators / factors in f32 (arithmetic stays f64)
   int fused = 1;               // one launch per Gauss-Seidel half sweep (gs_fused_kernel) where the level allows it
   int tail_from = -1;          // levels >= tail_from run in ONE single-workgroup launch (mltail.hip); -1 = none
   int gamma_from = 0, gamma_to = 0;   // levels [from, to) apply the coarse-grid correction twice (W-cycle there)
   double omega = 1.1;          // weight of the coarse-grid correction
   const nkp_tuning *tune = nullptr;   // the owning solver's knobs (set by ml_setup)
   size_t device_bytes = 0;
   double setup_seconds = 0.0;  // wall time of ml_setup
   int levels_on_device = 0;    // levels whose operator was built by the kernels of mlsetup.hip
   int batch_K = 0;             // the level vectors of ml_apply_batch exist for this many right-hand sides
};

// returns 0, or a negative nkp error code with a message in err
int ml_setup (MlHierarchy &H, int64_t n, const int *rowptr, const int *colind, const double *val,
              const int *blk_start, int64_t nblk, const int *col_i, const int *col_j, const int *col_t /* tracer of every column, or NULL = positional */,
              int tracer_cnt, int max_levels, int nu, int coarsest_rows, int verbose, int rank,
              hipStream_t st, char *err, size_t errlen, const nkp_tuning &tune, const CsrDev *A_dev = nullptr /* device copy of the same matrix, if the caller has one */);
void ml_free (MlHierarchy &H);
// z = V-cycle(r) in the ORIGINAL row order
void ml_apply (MlHierarchy &H, const double *r, double *z, hipStream_t st);
// the same cycle on K interleaved right-hand sides (r, z: n * K doubles, element (row, k) at row * K + k; K = 2 or 4); every
// column gets the bits ml_apply gives it alone.  ml_batch_prepare allocates the level vectors (0, or -2 = out of device memory).
int ml_batch_prepare (MlHierarchy &H, int K);
void ml_apply_batch (MlHierarchy &H, int K, const double *r, double *z, hipStream_t st);
void ml_apply_batch_split (MlHierarchy &H, int K, const double *const *src, double *z, double *const *dst, hipStream_t st);
// measurement helpers: launch one piece of a level-0 half sweep (0 residual rows, 1 column solves); compulsory HBM bytes of
// that piece or of the whole cycle (2)
void ml_time_piece (MlHierarchy &H, int which, hipStream_t st);
int64_t ml_bytes (const MlHierarchy &H, int which);
// the sub-cycle of the levels >= l0 in one single-workgroup launch (mltail.hip); 0 = launched
int ml_tail_launch (MlHierarchy &H, int l0, hipStream_t st);
