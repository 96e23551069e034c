// Internal device-side interfaces of libnkp_hip (gfx950 only).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nkp.h"

// the tuning a launcher uses when its object carries none: the plain defaults (no environment)
const nkp_tuning &nkp_builtin_tuning ();

#define NKP_LDSRES_CH 16         // substitution steps per factor chunk of colblock_apply_ldsres_kernel (group lengths are padded to it)
#define NKP_WAVE 64
#define NKP_MAX_K 512           // most basis vectors a fused update kernel takes (LDS coefficients)
#define NKP_SPMV_LDS_NNZ 2048    // CSR-stream row block: entries staged in LDS ...
#define NKP_SPMV_MAX_ROWS 256    // ... and rows (= threads of the SpMV workgroup) at most

// ---------------------------------------------------------------- CSR matrix on the device
struct CsrDev {
   int64_t n = 0, nnz = 0;
   int *rowptr = nullptr;      // [n+1]
   int *colind = nullptr;      // [nnz]
   double *val = nullptr;      // [nnz]
   float *valf = nullptr;      // [nnz] optional f32 copy of val (preconditioner operators: f32 storage, f64 arithmetic)
   // CSR-stream row blocks: block b owns rows [rowblk[b], rowblk[b+1]) whose entries fit the
   // LDS staging buffer (or a single long row)
   int *rowblk = nullptr;      // [nrowblk+1]
   int nrowblk = 0;
   // optional 2-byte column codes (internal format; the int32 colind stays for the setup kernels and
   // as the fallback): entry e of row block b holds (row - rowblk[b]) | id << 8 and its column is
   // row + dict[dict_ptr[b] + id], the block's dictionary of distinct (column - row) offsets.
   // dict_ptr[b+1] == dict_ptr[b] marks a block whose dictionary would exceed 256 entries (plain path).
   unsigned short *codes = nullptr;   // [nnz]
   int *dict = nullptr;
   int *dict_ptr = nullptr;           // [nrowblk+1]
   const nkp_tuning *tune = nullptr;  // launch shape knobs of the owning solver (NULL: built-in defaults)
};

// host helper: build the codes for a CSR matrix and its row blocks; returns the fraction of entries coded
double build_spmv_codes_host (int64_t n, const int *rowptr, const int *colind, const int *rowblk, int nrowblk,
                              unsigned short **codes_out, int **dict_out, int *ndict_out, int **dict_ptr_out);
// upload them into A (device); 0 = ok
int attach_spmv_codes (CsrDev &A, const int *h_rowptr, const int *h_colind, const int *h_rowblk, size_t *device_bytes);

// y = A x (mode 0) or y = b - A x (mode 1)
void launch_csr_spmv (const CsrDev &A, const double *x, double *y, const double *b, int mode, hipStream_t st);
// rows of the row blocks [rb0, rb1) only: y_rows = b_rows - (A x)_rows  (Gauss-Seidel colour sweep)
void launch_csr_residual_range (const CsrDev &A, int rb0, int rb1, const double *x, const double *b, double *y, hipStream_t st);
// rows of the row blocks [rb0, rb1) only, any mode (0: y = A x, 1: y = b - A x, 2: y = |A||x| + |b|)
void launch_csr_spmv_range (const CsrDev &A, int rb0, int rb1, const double *x, double *y, const double *b, int mode, hipStream_t st);
// y = |A| |x| + |b|   (denominator of the componentwise backward error)
void launch_csr_abs_spmv (const CsrDev &A, const double *x, const double *b, double *y, hipStream_t st);
// host helper: greedy row-block partition (host arrays)
void build_rowblocks_host (int64_t n, const int *rowptr, int **rowblk_out, int *nrowblk_out);

// ---------------------------------------------------------------- water-column blocks
// Banded LU (no pivoting) of every diagonal block, half-bandwidth P in {1,2,4}; SoA by
// diagonal: fac[(d+P)*n + row] holds, for d<0 the L multiplier l(row,row+d), for d=0 the
// RECIPROCAL of the U diagonal, for d>0 u(row,row+d).
struct ColBlocksDev {
   int64_t n = 0;
   int nblk = 0;
   int *blk_start = nullptr;   // [nblk+1]
   int P = 0;                  // half bandwidth actually stored
   int max_len = 0;            // longest block
   int dropped = 0;            // 1 if in-block entries beyond the band were dropped
   double *fac = nullptr;      // [(2P+1)*n]
   // lane-per-column layout for the apply kernel: groups of <= 64 consecutive blocks; group g's
   // factors live at fac_t + grp_base[g] as [(2P+1)][grp_maxlen[g]][64] (coalesced per step k)
   int ngrp = 0;
   int *grp_b0 = nullptr;      // [ngrp] first block of the group
   int *grp_nb = nullptr;      // [ngrp] blocks in the group (<= 64)
   int *grp_maxlen = nullptr;  // [ngrp] longest block of the group
   long long *grp_base = nullptr;   // [ngrp] offset into fac_t (doubles)
   int *grp_row0 = nullptr;    // [ngrp] first row of the group, [ngrp..2ngrp) its row count (no dependent blk_start lookups)
   int *col_slot = nullptr;    // [ngrp*gw] row offset of lane's column inside the group, [ngrp*gw..) its length
   double *fac_t = nullptr;
   float *fac_tf = nullptr;    // same layout in f32 (preconditioner: f32 storage, f64 arithmetic)
   int lds_doubles = 0;        // LDS staging need of the largest group (padded)
   int gw = 64;                // columns (lanes in use) per group
   int rhs_slots = 0;          // LDS doubles reserved for the staged right-hand side
   // fused Gauss-Seidel half sweep (gs_fused_kernel): row blocks of every group's rows
   int stream = 0;             // 1: 64 columns per wave, factors read straight from HBM (colblock_apply_stream_kernel)
   int ldsres = 0;             // 1: 32 columns per wave, factors streamed, the column resident in LDS (colblock_apply_ldsres_kernel);
                               // 2: the same with the factors packed four steps to a load and a static prefetch schedule (colblock_apply_ldspack_kernel)
   int gs_ok = 0;              // 1 if the level can run it (no row longer than GS_NNZ, LDS need within 64 KB)
   int gs_lds_bytes = 0;
   int *gs_rb_ptr = nullptr;   // [ngrp+1] first row-block boundary of the group
   int *gs_rb = nullptr;       // row-block boundaries (rows), groups back to back
   int *wave_desc = nullptr;   // levels run by gs_wave_kernel: per column {first row, rows, first entry, entries} -- one load instead of two dependent ones
   const nkp_tuning *tune = nullptr;   // kernel selection knobs of the owning solver (NULL: built-in defaults)
};

#define GS_THREADS 256
#define GS_NNZ 2048

// Build the lane-per-column layout.  ranges: nranges+1 block offsets; a group never straddles a
// range boundary (Gauss-Seidel colours).  grp_first[r] = first group of range r (nranges+1 out).
// Returns 0 or a HIP error code cast to int.
int colblock_build_lane_layout (ColBlocksDev &B, const int *h_blk_start, const int *ranges, int nranges,
                                int *grp_first, size_t *device_bytes, hipStream_t st, int f32 = 0, const int *h_rowptr = nullptr);
// one Gauss-Seidel half sweep over the groups [g0, g1) of one colour in ONE launch: r = b - L x on the groups' rows
// (x rows < split from xa, the others from xb), column solves, xout_rows = x_rows + z.  Returns non-zero (and does
// nothing) when the level cannot run the fused kernel.
int launch_gs_fused (const CsrDev &L, const ColBlocksDev &B, int g0, int g1, const double *xa, const double *xb, int split,
                     const double *b, double *xout, hipStream_t st);
// groups [g0, g1): z (+)= M^-1 r, one water column per LANE, rhs staged through LDS
void launch_colblock_apply_lanes (const ColBlocksDev &B, int g0, int g1, const double *r, double *z, int accumulate, hipStream_t st);

// max over blocks of the in-block half bandwidth and of the block length (device reduction)
void launch_colblock_measure (const CsrDev &A, const ColBlocksDev &B, int *d_out2 /* [bw, zero_diag_rows] */, hipStream_t st);
// extract + factor; *d_status receives the first row with a (near-)zero pivot + 1, else 0
void launch_colblock_factor (const CsrDev &A, ColBlocksDev &B, int *d_status, hipStream_t st);
// z = M^-1 r
void launch_colblock_apply (const ColBlocksDev &B, const double *r, double *z, hipStream_t st);
// blocks [b0, b1) only; accumulate: z_blk += M_blk^-1 r_blk, else z_blk = M_blk^-1 r_blk
void launch_colblock_apply_range (const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, hipStream_t st);
void launch_colblock_apply_range_r32 (const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, hipStream_t st);
// blocks [b0, b1) of one colour in ONE launch, one column per wave: xout_rows = x_rows + M_blk^-1 (b - L x)_rows, x taken from
// xa (rows < split) and xb (the others); r32 as above
void launch_build_wave_desc (const CsrDev &L, ColBlocksDev &B, hipStream_t st);   // fills B.wave_desc (allocated by the caller: 4 ints per column)
void launch_gs_wave (const CsrDev &L, const ColBlocksDev &B, int b0, int b1, const double *xa, const double *xb, int split, const double *b, double *xout,
                     int r32, hipStream_t st);

// ---------------------------------------------------------------- BLAS-1 style kernels
#define NKP_BATCH_MAX 8            // right-hand sides per sweep at most (interleave widths 2, 4, 8)
#define NKP_RED_BLOCKS 1024        // partial sums per reduction (fixed => deterministic)
#define NKP_DOT_CHUNK 8

// partial[(chunk*NKP_RED_BLOCKS + blk)*8 + c] ; then finish sums over blk in fixed order
// out[j] = sum_i V[j*ld+i] * w[i]  j<k ;  out[k] = sum_i w[i]^2
void launch_multi_dot (const void *V, int v_f32, int64_t ld, int k, const double *w, int64_t n, double *partial, double *out, hipStream_t st);
// w -= sum_j h[j] V_j (j<k);  out_nrm2[0] = ||w_new||^2 (via partial, deterministic)
void launch_update_w (const void *V, int v_f32, int64_t ld, int k, const double *h, double *w, int64_t n, double *partial, double *out_nrm2, hipStream_t st);
// y = alpha[0] * x   (alpha on device)
void launch_scale_to (const double *x, const double *alpha_dev, double *y, float *yf, int64_t n, hipStream_t st);
// x += sum_j c[j] Z_j (j<k)   (c on device)
void launch_axpy_multi (const double *Z, int64_t ld, int k, const double *c, double *x, int64_t n, hipStream_t st);
// out[0] = sum x_i*y_i
void launch_dot (const double *x, const double *y, int64_t n, double *partial, double *out, hipStream_t st);
// out[0] = max_i |r_i| / den_i  (den_i == 0 -> ignored when r_i == 0)
void launch_berr (const double *r, const double *den, int64_t n, double *partial, double *out, hipStream_t st);
// small device-side scalar programs of the Krylov drivers
// h[j] += h2[j] (j<k); h[k] = sqrt(nrm2); inv[0] = 1/h[k] (0 if h[k]==0)
void launch_finish_column (double *h, const double *h2, int k, const double *nrm2, double *inv, hipStream_t st);
void launch_finish_column_pythagoras (double *h, int k, double *inv, hipStream_t st);
// y = a*x + b*y style helpers for BiCGStab
void launch_axpby (double a, const double *x, double b, double *y, int64_t n, hipStream_t st);
void launch_copy (const double *x, double *y, int64_t n, hipStream_t st);
// y[i] = x[i] * w[i]   (y may alias x)
void launch_vmul (const double *x, const double *w, double *y, int64_t n, hipStream_t st);
void launch_fill (double *y, double v, int64_t n, hipStream_t st);

// ---------------------------------------------------------------- grid transfer / permutation
// coarse[I] = sum_{q in [rptr[I], rptr[I+1])} fine[ridx[q]]   (restriction = P^T, fixed order)
void launch_restrict_sum (const int *rptr, const int *ridx, const double *fine, double *coarse, int64_t nc, hipStream_t st);
// fine[i] += coarse[cmap[i]]                                  (prolongation = P)
void launch_prolong_add (const int *cmap, const double *coarse, double *fine, int64_t nf, double omega, hipStream_t st);
// out[i] = in[perm[i]]
void launch_gather (const int *perm, const double *in, double *out, int64_t n, hipStream_t st);
// out[perm[i]] = in[i]
void launch_scatter (const int *perm, const double *in, double *out, int64_t n, hipStream_t st);
// y = Minv x, dense row-major n x n (coarsest level)
void launch_dense_matvec (const double *Minv, const double *x, double *y, int n, hipStream_t st);
void launch_dense_matvec_f32 (const float *Minv, int ld, const double *x, double *y, int n, hipStream_t st);
void launch_dense_matvec_f32_batch (int K, const float *Minv, int ld, const double *x, double *y, int n, hipStream_t st);
int dense_inverse_blocked_device (int n, const int *h_rowptr, const int *h_col, const double *h_val, double **inv_out, float **invf_out, int *ldf_out,
                                  size_t *bytes, hipStream_t st);

// ---------------------------------------------------------------- K interleaved right-hand sides (batch.hip; X[i * K + k], K = 2 or 4)
void launch_interleave (int K, const double *const *src /* K pointers, NULL = zeros */, double *X, int64_t n, hipStream_t st);
void launch_deinterleave (int K, const double *X, double *const *dst /* K pointers, NULL = skip */, int64_t n, hipStream_t st);
// row blocks [rb0, rb1) of A: y = A x (mode 0) or y = b - A x (mode 1) on K columns
void launch_csr_spmv_batch (int K, const CsrDev &A, int rb0, int rb1, const double *x, double *y, const double *b, int mode, hipStream_t st);
void launch_restrict_sum_batch (int K, const int *rptr, const int *ridx, const double *fine, double *coarse, int64_t nc, hipStream_t st);
void launch_prolong_add_batch (int K, const int *cmap, const double *coarse, double *fine, int64_t nf, double omega, hipStream_t st);
void launch_gather_batch (int K, const int *perm, const double *in, double *out, int64_t n, hipStream_t st);
void launch_scatter_batch (int K, const int *perm, const double *in, double *out, int64_t n, hipStream_t st);
void launch_gather_interleave (int K, const int *perm, const double *const *src, double *out, int64_t n, hipStream_t st);
void launch_scatter_split (int K, const int *perm, const double *in, double *z, double *const *dst, int64_t n, hipStream_t st);
void launch_csr_spmv_batch_split (int K, const CsrDev &A, const double *x, double *const *dst, hipStream_t st);
void launch_dense_matvec_batch (int K, const double *Minv, const double *x, double *y, int n, hipStream_t st);
// water-column solves of blocks [b0, b1), one column per wave / the fused half sweep, K columns
void launch_colblock_apply_wave_batch (int K, const ColBlocksDev &B, int b0, int b1, const double *r, double *z, int accumulate, int r32, hipStream_t st);
void launch_gs_wave_batch (int K, const CsrDev &L, const ColBlocksDev &B, int b0, int b1, const double *xa, const double *xb, int split, const double *b, double *xout,
                           int r32, hipStream_t st);
// groups [g0, g1) with the packed lane layout (colblock.hip); non-zero = layout not served, use the wave kernel
int launch_colblock_apply_lanes_batch (int K, const ColBlocksDev &B, int g0, int g1, const double *r, double *z, int accumulate, hipStream_t st);
