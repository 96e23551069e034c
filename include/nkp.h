/* nkp.h -- C ABI of the MI355X-native sparse solve path (libnkp_hip.so).
 *
 * This is the drop-in boundary for the one thing the reference delegates to SuperLU_DIST:
 * "factor A once, then solve A x = b for each tracer right-hand side".  Every entry point
 * names the reference call site it replaces (paths relative to /root/reference):
 *
 *   nkp_create        <- dCreate_CompCol_Matrix_dist + set_default_options_dist +
 *                        ScalePermstructInit + LUstructInit + pdgssvx_ABglobal(nrhs=0)
 *                        src/solve_ABglobal.c:327-353  (factor-only call :353)
 *   nkp_create_dist   <- dCreate_CompRowLoc_Matrix_dist + pdgssvx(nrhs=0)
 *                        src/solve_ABdist.c:482-483, 518 (row block m_loc/fst_row :141-144)
 *   nkp_solve         <- pdgssvx_ABglobal(options.Fact=FACTORED, nrhs=1): B in, X out in the
 *                        same buffer, berr out, info as return code
 *                        src/solve_ABglobal.c:363, 393-395;  src/solve_ABdist.c:571
 *   nkp_destroy       <- Destroy_CompCol_Matrix_dist, Destroy_LU, ScalePermstructFree,
 *                        LUstructFree, superlu_gridexit   src/solve_ABglobal.c:412-424
 *   nkp_spmv          <- pdgsmv_AXglobal, the SpMV inside SuperLU's refinement loop
 *                        src/SuperLU_brief_tree.txt:21-22
 *   nkp_precond_apply <- pdgstrs_Bglobal (the triangular-solve phase)
 *                        src/SuperLU_brief_tree.txt:17
 *
 * Conventions (mirroring the reference): 0 = success, non-zero = failure; diagnostics go to
 * stderr prefixed "(rank)"; all indices 0-based; CSR with sorted, duplicate-free rows
 * (src/matrix.c:3826-3832).  Unlike SuperLU (which takes ownership of the arrays and frees
 * them in Destroy_*_Matrix_dist) nkp_create COPIES the caller's arrays to the device and
 * never frees or keeps host pointers.
 *
 * Plain C types only; no torch / HIP types in any signature.  Device pointers are passed
 * as void* and must belong to the device the solver was created on.
 */
#ifndef NKP_H
#define NKP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NKP_VERSION 1

typedef struct nkp_solver nkp_solver;

enum nkp_precond {
   NKP_PRECOND_NONE = 0,
   NKP_PRECOND_COLUMN_JACOBI = 1,   /* exact solve of every water-column block (= column-ILU(0):
                                       zero fill outside the column's band)                   */
   NKP_PRECOND_MULTILEVEL = 3       /* column blocks as smoother inside an aggregation hierarchy */
};

enum nkp_krylov { NKP_KRYLOV_FGMRES = 0, NKP_KRYLOV_BICGSTAB = 1 };

/* return codes of nkp_solve (SuperLU's `info` analogue) */
enum {
   NKP_OK = 0,
   NKP_NOT_CONVERGED = 1,    /* max_iters reached; x holds the best iterate, NOT written by the CLIs */
   NKP_BREAKDOWN = 2,
   NKP_OK_BERR = 3,          /* ||b-Ax||/||b|| stopped above rtol at the attainable f64 accuracy, but the componentwise
                                backward error is <= max (1e-14, rtol/100); x is written; callers decide (the CLIs accept it
                                only with NKP_ACCEPT_BERR=1 in the environment) */
   NKP_EINVAL = -1,
   NKP_ENOMEM = -2,
   NKP_EDEVICE = -3,         /* HIP runtime failure / no gfx950 device */
   NKP_ESINGULAR = -4,       /* zero pivot inside a water-column block */
   NKP_ECOMM = -5
};

/* Tuning knobs of the solve path (all optional).  The defaults are the measured best (DESIGN.md sections 2, 4, 11); everything
 * else exists for A/B runs and tests.  A solver reads them ONCE, in nkp_create: from nkp_options.tuning when the caller sets
 * it, else from nkp_default_tuning, which starts from the defaults and applies the NKP_* environment variables named on the
 * right (how the reference-shaped executables, which have no room for options, are steered).  They are per solver: nothing in
 * a solve, a cycle or a kernel launch looks at the environment. */
typedef struct nkp_tuning {
   int struct_size;          /* = sizeof(nkp_tuning), ABI guard */
   /* ---- hierarchy construction */
   int ml_split;             /* NKP_ML_SPLIT (1): connectivity-aware coarse cells inside the geometric groups */
   int ml_pocket;            /* NKP_ML_POCKET (4): same-depth connected sets of at most that many cells become one coarse cell */
   int ml_big_from;          /* NKP_ML_BIG_FROM (-3 = automatic): 4 x 4 groups from that level on, -1 = never */
   int ml_coarsest_rows;     /* NKP_ML_COARSEST_ROWS (8000): stop coarsening at that many rows (a dense level costs its bytes, an iterated one ~200 us of launch latencies) */
   int ml_dense_max;         /* NKP_ML_DENSE_MAX (8192): largest last level solved with a dense inverse */
   double ml_theta;          /* NKP_ML_THETA (0): edge threshold of the connectivity test */
   double ml_tau;            /* NKP_ML_TAU (0.01): leaf-stub threshold */
   int64_t ml_device_min;    /* NKP_ML_DEVICE_MIN (100000): levels with at least that many rows are built by the setup kernels,
                                smaller ones on the host (same hierarchy either way); < 0 = host only */
   /* ---- cycle */
   int ml_smooth_coarse;     /* NKP_ML_SMOOTH_COARSE (0 = like the fine levels): sweeps on levels >= ml_coarse_from */
   int ml_coarse_from;       /* NKP_ML_COARSE_FROM (2) */
   int ml_gamma_from, ml_gamma_to;   /* NKP_ML_GAMMA_FROM / _TO (0, 0): levels [from, to) visit the coarse level twice */
   int ml_f32;               /* NKP_ML_F32 (1): level operators and column factors stored in f32 (arithmetic in f64) */
   int ml_host_inverse;      /* NKP_ML_HOST_INVERSE (0): dense inverse of the last level on the host */
   int ml_fused;             /* NKP_ML_FUSED (0): one launch per Gauss-Seidel half sweep */
   int ml_fused_max_cols;    /* NKP_ML_FUSED_MAX_COLS (0 = all): ... only on levels with at most that many columns */
   int ml_wave_fused;        /* NKP_ML_WAVE_FUSED (1): levels solved one column per wave run each half sweep as ONE launch;
                                0 = never, N > 1 = only levels with at most N columns */
   int ml_coarsest_sweeps;   /* NKP_ML_COARSEST_SWEEPS (30): sweeps on a last level too large for a dense inverse */
   int64_t ml_tail_rows;     /* NKP_ML_TAIL_ROWS (0): the last levels with at most that many rows in one launch */
   double ml_omega;          /* NKP_ML_OMEGA (1.1): weight of the coarse-grid correction */
   /* ---- water-column solves: which kernel serves which level */
   int col_ldsres;           /* NKP_COL_LDSRES (2) */
   int col_stream;           /* NKP_COLSTREAM (1) */
   int col_stream_min;       /* NKP_COLSTREAM_MIN (-1 = per-kernel defaults) */
   int col_stream_gw;        /* NKP_COLSTREAM_GW (32) */
   int col_wave_max;         /* NKP_COLWAVE_MAX (8192): levels with at most that many columns solve one column per wave */
   int col_w3;               /* NKP_COL_W3 (1) */
   int col_group;            /* NKP_COLGROUP (8) */
   int col_pipe_min;         /* NKP_COLPIPE_MIN (0 = off) */
   int col_ldsres_early;     /* NKP_LDSRES_EARLY (0) */
   int col_ldsres_packed;    /* NKP_COL_PACKED (1): LDS-resident column kernel with factors packed four steps to a 16-byte load */
   /* ---- CSR SpMV launch shape */
   int spmv_variant;         /* NKP_SPMV_VARIANT (4) */
   int spmv_compress;        /* NKP_SPMV_COMPRESS (0): 2-byte column codes */
   int spmv_pipe_min;        /* NKP_SPMV_PIPE_MIN (1024): fewest row blocks for the pipelined kernel */
   int spmv_run;             /* NKP_SPMV_RUN (1): row blocks one workgroup walks */
   int spmv_wgs;             /* NKP_SPMV_WGS (256): workgroups per CU at most */
   /* ---- Krylov / distributed flavour / setup */
   int rhs_batch;            /* NKP_RHS_BATCH (1): several right-hand sides of one call share the sweeps over the matrix and the
                                hierarchy (nkp_solve_batch_device; same bits per column); 0 = one at a time, 1 or 4 = groups of up to
                                four, 2 = pairs, 8 = groups of up to eight (measured at 1 degree: no better per solve than four -- with
                                four vectors interleaved the vectors, not the matrix, are most of every kernel's bytes) */
   int precond_steps;        /* NKP_PRECOND_STEPS (0 = leave nkp_options.precond_steps) */
   int equil;                /* NKP_EQUIL (-1 = leave nkp_options.equil) */
   int dist_overlap;         /* NKP_DIST_OVERLAP (1): halo exchange behind the interior rows */
   int dist_ras;             /* NKP_DIST_RAS (1): one ring of the neighbours' columns in the rank's hierarchy */
   int force_dist;           /* NKP_FORCE_DIST (0): distributed code path with one rank */
   int setup_threads;        /* NKP_SETUP_THREADS (0 = automatic): host threads of the setup loops */
   int plan_times;           /* NKP_ML_PLAN_TIMES (0): print the split of the host-side aggregation */
   int ml_drop_intertracer;  /* NKP_ML_DROP_INTERTRACER (0): developer switch, hierarchy without inter-tracer couplings */
   int dist_one_reduce;      /* NKP_DIST_ONE_REDUCE (0): distributed Arnoldi step with ONE allreduce -- the norm of the orthogonalised
                                vector comes from the reduced multi-dot message (w.w - sum h^2) instead of a second allreduce.
                                Off by default: the identity assumes an orthonormal basis, which one Gram-Schmidt pass keeps only
                                to 1e-6 or so, and the solve pays for it (2 ranks, 40x46x20: 48 iterations against 45) -- more than
                                the 20-30 us allreduce it saves per 3 ms step */
   int ml_huge_from;         /* NKP_ML_HUGE_FROM (-1 = never): 8 x 8 groups from that level on */
   int col_ldsres_min;       /* NKP_COL_LDSRES_MIN (0 = automatic): fewest columns of a level served by the LDS-resident column kernels */
   int col_sort_groups;      /* NKP_COL_SORT_GROUPS (1): packed column layout groups a colour's columns by length (less zero padding) */
   int batch_spmv_rows;      /* NKP_BATCH_SPMV_ROWS (1): batched SpMV stages the (value, column) stream in LDS and lets each row's lane gather
                                its own K-wide rows of x; 0 = products parked in LDS, K / 2 passes */
} nkp_tuning;

/* defaults, then the NKP_* environment overrides listed above */
int nkp_default_tuning (nkp_tuning *t);

typedef struct nkp_options {
   int struct_size;      /* = sizeof(nkp_options), ABI guard                                  */
   int precond;          /* enum nkp_precond                                                  */
   int krylov;           /* enum nkp_krylov                                                   */
   int restart;          /* FGMRES restart length m                                           */
   int max_iters;        /* total Krylov iterations allowed per right-hand side               */
   double rtol;          /* stop when ||b - A x||_2 <= rtol * ||b||_2 (true residual)         */
   double atol;          /* ... or <= atol                                                    */
   int device;           /* HIP device ordinal; -1 = leave the current device                 */
   int verbose;          /* the reference's dbg_lvl: 0 silent, 1 progress, 2 per-iteration    */
   int rank;             /* printed as the "(rank)" message prefix (reference `iam`)          */
   int reorth;           /* 0 = one classical Gram-Schmidt pass (default), 1 = two passes     */
   int ml_levels;        /* multilevel: max levels (0 = automatic)                            */
   int ml_smooth;        /* multilevel: smoothing sweeps per level per half-cycle             */
   int basis_f32;        /* 1: store the Krylov basis V in f32 for the Gram-Schmidt passes (the solution update uses
                            the f64 Z vectors, the true residual is recomputed in f64 at every restart).  Default 0:
                            with a single Gram-Schmidt pass the f32 basis can double the iteration count. */
   int precond_steps;    /* preconditioner cycles per Krylov iteration, chained by defect correction against A:
                            z = M r; z += M (r - A z); ...  0 = automatic (= 1 since round 2) */
   int equil;            /* row equilibration (SuperLU's Equil=YES, reference src/solve_ABglobal.c:332): FGMRES minimises
                            ||R (b - A x)||_2 with R = diag (1 / max_j |a_ij|) instead of ||b - A x||_2; the stopping test
                            stays on the unscaled residual.  0 = automatic (off unless NKP_EQUIL=1), 1 = on, -1 = off.
                            Column scaling has no effect on a right-preconditioned iteration and is not applied. */
   int reserved[4];
   /* multilevel, optional: grid position (i, j) of every water-column block, nblk entries each
    * (tracer_state_ind_to_i/_j at the block's first row, reference src/matrix.c:322-329).  With
    * them columns are aggregated 2 x 2 in (i, j) and coloured (i + j) % 2; without them
    * (NULL) the setup falls back to pairwise matching on the column graph.  Host pointers, read
    * during nkp_create only. */
   const int32_t *col_i;
   const int32_t *col_j;
   /* optional: tracer of every water-column block (nblk entries).  NULL = tracer-major rows (reference
    * src/matrix.c:778-784): block c belongs to tracer c / (nblk / coupled_tracer_cnt).  Needed when the rows were
    * reordered cell-major (nkp_cell_major_order below).  Columns of different tracers are never aggregated together. */
   const int32_t *col_t;
   /* optional: tuning knobs (NULL = nkp_default_tuning: defaults + NKP_* environment).  Read during nkp_create only. */
   const nkp_tuning *tuning;
} nkp_options;

int nkp_default_options (nkp_options *opt);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int nkp_device_count (void);

/* Setup ("factor") -- host CSR in, device-resident solver out.
 *   blk_start[nblk+1]: row offsets of the water-column blocks (contiguous, ascending,
 *   blk_start[0]=0, blk_start[nblk]=n).  Rows of one block are the levels k=0..KMT-1 of one
 *   (tracer, column) (src/matrix.c:239-251, 778-784).  NULL => every row its own block
 *   (point Jacobi). */
int nkp_create (nkp_solver **out, const nkp_options *opt, int64_t n, int64_t nnz,
                const int32_t *rowptr, const int32_t *colind, const double *val,
                const int32_t *blk_start, int64_t nblk, int coupled_tracer_cnt);

/* Same with 64-bit row pointers (what a CDF-5 matrix file or a caller that counts entries in int64 holds).  Entry
 * offsets are stored in 32 bits on the device, so ONE GPU takes at most 2^31 - 1 entries (25 GB of values and column
 * indices); a larger system is row-partitioned with nkp_create_dist, where the limit applies to each rank's block --
 * the 0.25 degree x 4 tracer system (4.2 G entries, n = 203 M < 2^31) is representable on 4 or 8 ranks.  Returns
 * NKP_EINVAL with that explanation when rowptr[n] does not fit. */
int nkp_create64 (nkp_solver **out, const nkp_options *opt, int64_t n, const int64_t *rowptr /* n+1 */,
                  const int32_t *colind, const double *val, const int32_t *blk_start, int64_t nblk, int coupled_tracer_cnt);

/* Solve nrhs systems; b (host, column-major, leading dimension ldb >= n) is overwritten by x
 * when the return code is NKP_OK or NKP_NOT_CONVERGED.  berr[r] receives the componentwise
 * backward error max_i |b-Ax|_i / (|A||x|+|b|)_i like SuperLU's; iters/relres per rhs.
 * Any of berr/iters/relres may be NULL.
 * NKP_OK means ||b-Ax||/||b|| <= rtol (or <= atol absolute) on the true residual, nothing else.  When rounding
 * stops the residual above rtol (badly scaled rows: even a direct solve with refinement then stays above it) the solve
 * ends after a few restart cycles instead of running to max_iters and returns NKP_OK_BERR if the componentwise
 * backward error -- the accuracy measure the reference itself reports (berr, src/solve_ABglobal.c:396-398) -- is
 * <= max (1e-14, rtol / 100), else NKP_NOT_CONVERGED. */
int nkp_solve (nkp_solver *s, double *b_in_x_out, int nrhs, int64_t ldb,
               double *berr, int *iters, double *relres);

/* Several right-hand sides at once -- the reference's RHS loop (src/solve_ABglobal.c:370-409) with the tracers sharing every sweep
 * over the matrix and the hierarchy: d_B / d_X hold nrhs vectors of n doubles on the solver's device, vector c at offset c * ldb
 * (d_X may alias d_B).  Systems are solved in groups of up to four; each keeps its own FGMRES recurrence and stopping test, the
 * operator and preconditioner applications of a Krylov step are one pass for the group.  Column c of the result has the bits
 * nkp_solve_device gives for that right-hand side alone, iters[c] / relres[c] / berr[c] likewise; the return code is the worst of
 * the columns'.  Needs K - 1 more sets of work vectors (kept for later calls).  Falls back to one at a time where the batched path
 * does not apply (BiCGStab, row equilibration, chained cycles, the distributed flavour) or with nkp_tuning.rhs_batch = 0.
 * nkp_solve with nrhs >= 2 takes the same path. */
int nkp_solve_batch_device (nkp_solver *s, int nrhs, const void *d_B, void *d_X, int64_t ldb, double *berr, int *iters, double *relres);

/* Same, with b and x already resident on the solver's device (x may alias b); x_inout is also
 * the initial guess when use_guess != 0. */
int nkp_solve_device (nkp_solver *s, const void *d_b, void *d_x, int use_guess,
                      double *berr, int *iters, double *relres);

/* y = A x, exposed for parity and roofline tests (host and device flavours). */
int nkp_spmv (nkp_solver *s, const double *x, double *y);
int nkp_spmv_device (nkp_solver *s, const void *d_x, void *d_y);

/* z = M^-1 r with the configured preconditioner (host buffers). */
int nkp_precond_apply (nkp_solver *s, const double *r, double *z);

/* Deterministic device reductions used by the Krylov drivers, exposed for parity tests:
 * out[j] = sum_i V[j*ld + i] * w[i], j < k   (host buffers). */
int nkp_multi_dot (nkp_solver *s, const double *V, int64_t ld, int k, const double *w, double *out);

/* Average duration (ms) of `reps` back-to-back launches of one kernel on the solver's stream,
 * measured with HIP events on that stream.  which: 0 = CSR SpMV, 1 = preconditioner apply,
 * 2 = one full Krylov iteration body at restart position `arg` (0 <= arg < restart); multilevel only: 3 = the
 * smoother's residual rows of one colour of the fine level, 4 = the water-column solves of that colour. */
int nkp_time_kernel (nkp_solver *s, int which, int arg, int reps, double *avg_ms);

/* Introspection: key = "n", "nnz", "nblk", "band", "levels", "spmv_bytes", "device_bytes", "precond_steps", "equil";
 * "create_us" (wall time of nkp_create), "ml_setup_us" (of which: the hierarchy), "ml_levels_on_device" (levels whose operator
 * the setup kernels built; the smaller ones are built on the host);
 * compulsory HBM bytes of the pieces nkp_time_kernel times: "smoother_spmv_bytes", "column_solve_bytes", "cycle_bytes";
 * distributed flavour: "dist_overlap" (halo exchange hidden behind the interior rows), "dist_interior_rowblocks",
 * "dist_ras" (hierarchy overlaps the neighbouring ranks), "dist_ras_rows" (rows of other ranks in this rank's hierarchy). */
int64_t nkp_get_int (nkp_solver *s, const char *key);

/* Use an externally owned HIP stream (hipStream_t cast to void*) instead of the solver's own;
 * NULL = the device's default stream. */
int nkp_set_stream (nkp_solver *s, void *hip_stream);

void nkp_destroy (nkp_solver *s);

/* Message of the last failure on this thread ("" if none). */
const char *nkp_last_error (void);

/* ---- row-distributed flavour (one process per GPU, RCCL over xGMI) --------------------- */

/* What the distributed solver needs from the outside world: four collectives.  The library ships
 * an RCCL implementation (nkp_comm_rccl_init); a host program that already owns a communicator
 * (torch.distributed in bench.py, MPI in a port of src/solve_ABdist.c) passes its own callbacks.
 * All callbacks return 0 on success and are called collectively, in the same order, by every rank.
 * This is the stand-in for superlu_gridinit + SuperLU_DIST's internal MPI (src/solve_ABdist.c:461). */
typedef struct nkp_comm_ops {
   void *ctx;
   int rank, nranks;
   /* in-place reduction of `count` doubles in DEVICE memory; op 0 = sum, 1 = max; ordered on hip_stream */
   int (*allreduce) (void *ctx, void *dev_buf, int count, int op, void *hip_stream);
   /* personalised exchange of doubles in DEVICE memory: send_counts[p] values for rank p are taken
    * consecutively from dev_send, recv_counts[p] values from rank p land consecutively in dev_recv */
   int (*alltoallv) (void *ctx, const void *dev_send, const int *send_counts, void *dev_recv, const int *recv_counts, void *hip_stream);
   /* setup only, HOST memory: the same exchange for int32, and an allgather of one int64 per rank */
   int (*alltoallv_i32_host) (void *ctx, const int32_t *send, const int *send_counts, int32_t *recv, const int *recv_counts);
   int (*allgather_i64_host) (void *ctx, int64_t mine, int64_t *all /* nranks */);
} nkp_comm_ops;

/* 128-byte RCCL unique id, created on rank 0 and broadcast by the caller (torch.distributed,
 * MPI or a shared file). */
int nkp_comm_unique_id (void *id128);
/* Fill `ops` with the built-in RCCL implementation (ncclCommInitRank on the current device). */
int nkp_comm_rccl_init (nkp_comm_ops *ops, const void *id128, int rank, int nranks);
void nkp_comm_rccl_free (nkp_comm_ops *ops);
/* Fill `ops` with a host-staged transport over a directory every rank can write (one file per rank and collective).
 * Test-grade: it lets several processes that share ONE GPU run the distributed code path (RCCL needs one GPU per rank).
 * Every blocking read has a deadline (NKP_COMM_TIMEOUT seconds, default 120): a dead peer fails the collective. */
int nkp_comm_file_init (nkp_comm_ops *ops, const char *dir, int rank, int nranks);
void nkp_comm_file_free (nkp_comm_ops *ops);

/* Local row block [fst_row, fst_row + m_loc) with GLOBAL column indices, rowptr rebased to 0
 * -- exactly what dCreate_CompRowLoc_Matrix_dist receives (src/solve_ABdist.c:482-483).
 * blk_start_loc holds the local block offsets (relative to fst_row, blk_start_loc[nblk_loc] =
 * m_loc); a water column must not straddle ranks.  Collective over all ranks.
 * The Krylov iteration is global (halo exchange before every SpMV, one allreduce per
 * Gram-Schmidt pass); the multilevel preconditioner is one hierarchy per rank that, where the cut is
 * lateral (opt->col_i / col_j given), also covers one ring of the neighbouring ranks' water columns
 * (restricted additive Schwarz: setup fetches those rows from their owners through alltoallv_i32_host,
 * every application fetches their residual through alltoallv; NKP_DIST_RAS=0 = diagonal block only).  nkp_solve / nkp_solve_device then take and return the LOCAL slice
 * of b / x, like pdgssvx with ldb = m_loc (src/solve_ABdist.c:571). */
int nkp_create_dist (nkp_solver **out, const nkp_options *opt, int64_t n_global, int64_t fst_row,
                     int64_t m_loc, int64_t nnz_loc, const int32_t *rowptr_loc,
                     const int32_t *colind_glob, const double *val,
                     const int32_t *blk_start_loc, int64_t nblk_loc, int coupled_tracer_cnt,
                     const nkp_comm_ops *comm);

/* Cell-major ordering of a coupled system.  The reference stores coupled tracers tracer-major
 * (src/matrix.c:778-784): cutting such a system into contiguous row blocks (src/solve_ABdist.c:141-144) puts the
 * same-cell coupling entries (src/matrix.c:955-961) off-rank in EVERY row.  In cell-major order -- for every water-column
 * position the columns of all tracers one after the other -- the same contiguous blocks are latitude bands of the whole
 * coupled system, the couplings are rank-local and the halo is the band edge.  Host only, no GPU needed.
 *   nkp_cell_major_order: blk_start[nblk+1] = tracer-major block offsets, nblk = cnt * (blocks per tracer), every tracer
 *     with the same block lengths.  Out: perm[n] (new row -> old row), blk_start_new[nblk+1], col_t[nblk] (tracer of
 *     every new block), col_src[nblk] (old block of every new block: col_i_new[c] = col_i[col_src[c]]).
 *   nkp_permuted_rows: rows [r0, r1) of P A P^T (new numbering on both sides) from the tracer-major CSR; inv[n] = old row
 *     -> new row.  rowptr_loc[r1 - r0 + 1] rebased to 0; colind_loc / val_loc sized for the entries of those rows, columns
 *     sorted ascending within a row. */
int nkp_cell_major_order (int64_t nblk, const int32_t *blk_start, int cnt, int32_t *perm, int32_t *blk_start_new,
                          int32_t *col_t, int32_t *col_src);
int nkp_permuted_rows (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const int32_t *perm,
                       const int32_t *inv, int64_t r0, int64_t r1, int32_t *rowptr_loc, int32_t *colind_loc, double *val_loc);

/* Multi-RHS concurrency: the reference solves its right-hand sides one after the other
 * against one factorisation (RHS loop, src/solve_ABglobal.c:370-409).  nkp_clone gives a second set of work
 * vectors and a second stream on the SAME device-resident matrix, factors and hierarchy (nothing is copied), so
 * that several right-hand sides can be in flight at once: one clone per host thread, each calling nkp_solve /
 * nkp_solve_device on its own handle.  Results are bit-identical to solving on the original.  Destroy clones
 * (nkp_destroy) before the solver they were cloned from.  Single-GPU solvers only. */
int nkp_clone (nkp_solver *src, nkp_solver **out);

/* hipSetDevice for host programs that do not link HIP themselves (call before nkp_comm_rccl_init). */
int nkp_set_device (int device);

/* Collective: concatenate every rank's local slice (host, m_loc doubles) in rank order into
 * x_global (host, n_global doubles, significant on rank 0 only) -- what put_B_dist does with
 * MPI_Send/MPI_Recv tag 4 (src/solve_ABdist.c:377, 406). */
int nkp_gather_root (nkp_solver *s, const double *x_loc, double *x_global);

/* The whole host-side plan of nkp_create_dist as an object, for tests without a GPU: collective over the ranks through the
 * HOST callbacks of `comm` (alltoallv_i32_host, allgather_i64_host); no HIP call.  Fields (nkp_dist_plan_size gives the
 * element count, nkp_dist_plan_copy copies): SpMV side -- "colind_ext" (int32, the local columns renumbered [own | halo]),
 * "halo_rows" (global rows received before every SpMV; with grid positions in opt the halo is completed to whole water
 * columns), "send_rows" (own rows sent, grouped by destination), "need" / "give" (per-rank counts); hierarchy side, filled
 * when the overlap is on (nkp_dist_plan_size (p, "ras") == 1) -- the matrix on [own rows | overlap rows]: "rowptr",
 * "colind", "val" (double), "blk_start", "col_i", "col_j", "col_t", and "sel_hpos" (position in the halo of every overlap row). */
typedef struct nkp_dist_plan nkp_dist_plan;
int nkp_dist_overlap_plan_host (nkp_dist_plan **out, const nkp_options *opt, int64_t n_global, int64_t fst_row, int64_t m_loc,
                                int64_t nnz_loc, const int32_t *rowptr_loc, const int32_t *colind_glob, const double *val,
                                const int32_t *blk_start_loc, int64_t nblk_loc, int coupled_tracer_cnt, const nkp_comm_ops *comm);
int64_t nkp_dist_plan_size (const nkp_dist_plan *p, const char *what);
int nkp_dist_plan_copy (const nkp_dist_plan *p, const char *what, void *dst);
void nkp_dist_plan_free (nkp_dist_plan *p);

/* Host-only planning step of nkp_create_dist, exposed so the partition / halo logic can be tested
 * without a GPU: given this rank's rows and the row offsets of all ranks (starts[nranks+1]),
 * writes the remapped column indices (local rows -> [0, m_loc), halo -> m_loc + position in the
 * sorted list of needed off-rank rows) into colind_ext[nnz_loc], the needed global rows into
 * halo_rows (capacity nnz_loc; *n_halo entries used) and how many come from each rank into
 * need_counts[nranks].  Returns 0 or NKP_EINVAL. */
int nkp_dist_plan_host (int64_t m_loc, int64_t nnz_loc, const int32_t *rowptr_loc, const int32_t *colind_glob,
                        int rank, int nranks, const int64_t *starts, int32_t *colind_ext, int32_t *halo_rows,
                        int64_t *n_halo, int32_t *need_counts);

/* Introspection of the multilevel hierarchy as it sits on the device (tests: the levels the setup kernels build must equal
 * the ones the host routines build, entry for entry).  Copies one array of level `level` (0 = finest) to dst and returns its
 * element count (dst == NULL: the count only); negative = error.  what: "rowptr" (int32, rows + 1), "colind" (int32),
 * "valf" (float, the f32 storage of a level operator) / "val" (double, where the f64 values are kept), "cmap" (int32: row ->
 * row of the next level), "rptr" / "ridx" (int32: row of the next level -> its rows here), "blk_start" (int32), "fac"
 * (double, band factors of the column blocks), "perm0" (int32, level 0: row -> original row), "coarse_inv" (double, last
 * level).  All in the level's colour-major row order. */
int64_t nkp_ml_level_array (nkp_solver *s, int level, const char *what, void *dst, int64_t capacity_bytes);

/* Host-only planning step of the multilevel preconditioner inside nkp_create, exposed so the aggregation logic can
 * be tested without a GPU: builds the low-order twin, the coarse cells of every level (geometric groups split by
 * lateral connectivity) and the Galerkin products, and reports
 *   *n_levels, rows[l] (l < *n_levels), and for every level l < *n_levels - 1, concatenated in level order:
 *   cmap   : coarse row (level l+1 numbering) of every row of level l            (sum of rows[0 .. n_levels-2] entries)
 *   col_of : column block of every row of level l+1                              (sum of rows[1 .. n_levels-1] entries)
 * `capacity` bounds both output arrays (entries).  col_i / col_j as in nkp_options.  Returns 0, NKP_EINVAL, or NKP_ENOMEM
 * when the capacity is too small. */
int nkp_ml_plan_host (int64_t n, const int32_t *rowptr, const int32_t *colind, const double *val, const int32_t *blk_start,
                      int64_t nblk, const int32_t *col_i, const int32_t *col_j, int coupled_tracer_cnt, int max_levels,
                      int coarsest_rows, int64_t capacity, int *n_levels, int64_t *rows, int32_t *cmap, int32_t *col_of);

#ifdef __cplusplus
}
#endif
#endif
