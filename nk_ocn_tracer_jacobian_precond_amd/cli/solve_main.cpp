// solve_ABglobal / solve_ABdist -- drop-in executables.
//
//   solve_AB* [-D dbg_lvl] [-n nprow[,npcol]] [-v vars] matrix_fname inout_fname
//
// Same command line, file contract, message prefix "(rank)" and exit codes as the reference
// mains (src/solve_ABglobal.c:37-99, 271-431; src/solve_ABdist.c:42-104, 422-612).  The
// SuperLU_DIST calls are replaced by the C ABI of include/nkp.h:
//
//   dCreate_*_Matrix_dist + pdgssvx*(nrhs=0)     -> nkp_create / nkp_create_dist  ("factor")
//   pdgssvx*(Fact=FACTORED, nrhs=1), B <- X      -> nkp_solve
//   Destroy_* / superlu_gridexit                 -> nkp_destroy
//
// Deliberate differences: -v is required (the reference dereferences
// NULL without it, solve_ABglobal.c:24,370); a solve that does not converge exits non-zero and
// does NOT overwrite the tracer variable (the reference ignores SuperLU's info, :395-405);
// -n is parsed as before but only sizes the rank count of the distributed flavour.
// Extra knobs come from the environment so legacy invocations keep working:
//   NKP_RTOL NKP_MAX_ITERS NKP_RESTART NKP_PRECOND(none|column|multilevel) NKP_KRYLOV(fgmres|bicgstab) NKP_ML_SMOOTH
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <algorithm>
#include <vector>
#include <time.h>
#include <unistd.h>

#include "../../include/nkp.h"
#include "../host/nkp_host.h"

#ifdef NKP_DIST
static const char *prog_solver = "nkp_create_dist";
#else
static const char *prog_solver = "nkp_create";
#endif

static long nprow, npcol;
static char *vars = NULL;
static char *matrix_fname = NULL;
static char *inout_fname = NULL;

static int parse_cmd_line (int argc, char **argv)
{
   const char *usage_msg = "usage: jacobian_precond [-D dbg_lvl] [-n nprow[,npcol]] [-v vars] matrix_fname inout_fname\n   vars: comma-separated variable names; NAME=SOLNAME reads the right-hand side from NAME and writes the solution to SOLNAME";
   int opt;

   while ((opt = getopt (argc, argv, "D:n:v:h")) != -1) {
      switch (opt) {
      case '?':
      case 'h':
         fprintf (stderr, "(%d) %s\n", iam, usage_msg);
         return 1;
      case 'D':
         if (parse_to_int (optarg, &dbg_lvl)) {
            fprintf (stderr, "(%d) error parsing argument '%s' for option '%c'\n", iam, optarg, opt);
            return 1;
         }
         break;
      case 'n': {
            char *first = strtok (optarg, ",");
            char *second = strtok (NULL, ",");
            if (parse_to_long (first, &nprow)) {
               fprintf (stderr, "(%d) error parsing argument '%s' for option '%c'\n", iam, first ? first : "", opt);
               return 1;
            }
            npcol = nprow;
            if (second && parse_to_long (second, &npcol)) {
               fprintf (stderr, "(%d) error parsing argument '%s' for option '%c'\n", iam, second, opt);
               return 1;
            }
            break;
         }
      case 'v':
         free (vars);
         if ((vars = strdup (optarg)) == NULL) {
            fprintf (stderr, "(%d) malloc failed in parse_cmd_line for vars\n", iam);
            return 1;
         }
         break;
      default:
         fprintf (stderr, "(%d) internal error: unhandled option '-%c'\n", iam, opt);
         return 1;
      }
   }
   if (optind != argc - 2) {
      fprintf (stderr, "(%d) unexpected number of arguments\n%s\n", iam, usage_msg);
      return 1;
   }
   matrix_fname = argv[optind++];
   inout_fname = argv[optind++];
   if (vars == NULL || vars[0] == '\0') {
      fprintf (stderr, "(%d) no variables given, nothing to solve (-v vars)\n%s\n", iam, usage_msg);
      return 1;
   }
   return 0;
}

static void trace (const char *what, const char *subname)
{
   if (dbg_lvl > 1) {
      printf ("(%d) %s %s\n", iam, what, subname);
      fflush (stdout);
   }
}

// "-v RHS=SOL": the right-hand side is read from variable RHS and the solution written into variable SOL (which must exist with the
// same shape; its land values are kept) -- the reference's TODO:1, "separate RHS and soln vectors in solve_AB".  A plain name is both.
static std::string rhs_name (const char *tok) { const char *eq = strchr (tok, '='); return eq ? std::string (tok, (size_t) (eq - tok)) : std::string (tok); }
static std::string sol_name (const char *tok) { const char *eq = strchr (tok, '='); return eq ? std::string (eq + 1) : std::string (tok); }

// read each tracer of the group as a [km][jmt][imt] cube and flatten it into B
// (reference get_B_global, src/solve_ABglobal.c:153-208)
static int get_B_global (char **vars_per_solve, double *B)
{
   const char *subname = "get_B_global";
   trace ("entering", subname);
   double ***field_3d = malloc_3d_double (km, jmt, imt);
   if (field_3d == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for field_3d\n", iam, subname);
      return 1;
   }
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      const std::string vname = rhs_name (vars_per_solve[t]);
      size_t nelems = 0;
      if (dbg_lvl)
         printf ("(%d) reading %s from %s\n", iam, vname.c_str (), inout_fname);
      if (nkp_var_nelems (inout_fname, (char *) vname.c_str (), &nelems))
         return 1;
      if (nelems != (size_t) km * (size_t) jmt * (size_t) imt) {
         fprintf (stderr, "(%d) %s: variable %s holds %zu values, expected km*jmt*imt = %zu\n", iam, subname, vname.c_str (), nelems,
                  (size_t) km * (size_t) jmt * (size_t) imt);
         return 1;
      }
      if (get_var_3d_double (inout_fname, (char *) vname.c_str (), field_3d))
         return 1;
      nkp_flatten_tracer (t, field_3d, B);
   }
   free_3d_double (field_3d);
   trace ("exiting", subname);
   return 0;
}

// re-read each cube (non-ocean values must survive), scatter X into it, write it back in place
// (reference put_B_global, src/solve_ABglobal.c:212-267)
static int put_B_global (char **vars_per_solve, double *B)
{
   const char *subname = "put_B_global";
   trace ("entering", subname);
   double ***field_3d = malloc_3d_double (km, jmt, imt);
   if (field_3d == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for field_3d\n", iam, subname);
      return 1;
   }
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      const std::string vname = sol_name (vars_per_solve[t]);
      if (get_var_3d_double (inout_fname, (char *) vname.c_str (), field_3d))
         return 1;
      nkp_unflatten_tracer (t, B, field_3d);
      if (dbg_lvl)
         printf ("(%d) writing %s to %s\n", iam, vname.c_str (), inout_fname);
      if (put_var_3d_double (inout_fname, (char *) vname.c_str (), field_3d))
         return 1;
   }
   free_3d_double (field_3d);
   trace ("exiting", subname);
   return 0;
}

static void options_from_env (nkp_options *o)
{
   const char *e;
   double d;
   int i;
   if ((e = getenv ("NKP_RTOL")) && !parse_to_double ((char *) e, &d)) o->rtol = d;
   if ((e = getenv ("NKP_MAX_ITERS")) && !parse_to_int ((char *) e, &i)) o->max_iters = i;
   if ((e = getenv ("NKP_RESTART")) && !parse_to_int ((char *) e, &i)) o->restart = i;
   if ((e = getenv ("NKP_ML_SMOOTH")) && !parse_to_int ((char *) e, &i)) o->ml_smooth = i;
   if ((e = getenv ("NKP_PRECOND"))) {
      if (!strcmp (e, "none")) o->precond = NKP_PRECOND_NONE;
      else if (!strcmp (e, "column")) o->precond = NKP_PRECOND_COLUMN_JACOBI;
      else if (!strcmp (e, "multilevel")) o->precond = NKP_PRECOND_MULTILEVEL;
      else fprintf (stderr, "(%d) ignoring unknown NKP_PRECOND '%s'\n", iam, e);
   }
   if ((e = getenv ("NKP_KRYLOV"))) {
      if (!strcmp (e, "fgmres")) o->krylov = NKP_KRYLOV_FGMRES;
      else if (!strcmp (e, "bicgstab")) o->krylov = NKP_KRYLOV_BICGSTAB;
      else fprintf (stderr, "(%d) ignoring unknown NKP_KRYLOV '%s'\n", iam, e);
   }
}

// NKP_OK_BERR: the residual stopped above the tolerance at the attainable f64 accuracy while the componentwise backward
// error -- the only accuracy figure the reference prints (src/solve_ABglobal.c:396-398) -- is at rounding level.  The
// reference would write such a result without looking; this program writes it only when the caller opts in.
static int berr_verdict (int info, double relres, double berr)
{
   if (info != NKP_OK_BERR) return info;
   const char *e = getenv ("NKP_ACCEPT_BERR");
   if (e && atoi (e) != 0) {
      printf ("(%d) residual %.3e above the tolerance at the attainable accuracy, backward error %.3e: accepted (NKP_ACCEPT_BERR)\n", iam, relres, berr);
      return 0;
   }
   return info;
}

int main (int argc, char *argv[])
{
   dbg_lvl = 0;
   nprow = npcol = 4;          // the reference's default process grid (solve_ABglobal.c:296)
   {
      const char *r = getenv ("RANK");
      int v;
      iam = (r && !parse_to_int ((char *) r, &v)) ? v : 0;
   }
   if (parse_cmd_line (argc, argv))
      exit (EXIT_FAILURE);
   int world = 1, local_rank = 0, use_comm = 0;
   (void) local_rank;
#ifdef NKP_DIST
   // one process per GPU, launched with RANK / WORLD_SIZE / LOCAL_RANK in the environment (mpirun, torchrun,
   // srun ... all set them or an equivalent); NKP_RCCL_ID_FILE names a file on a shared path through which
   // rank 0 hands the RCCL unique id to the others (the reference uses MPI_COMM_WORLD, src/solve_ABdist.c:461)
   {
      const char *e;
      int v;
      if ((e = getenv ("WORLD_SIZE")) && !parse_to_int ((char *) e, &v) && v > 0) world = v;
      local_rank = iam;
      if ((e = getenv ("LOCAL_RANK")) && !parse_to_int ((char *) e, &v) && v >= 0) local_rank = v;
   }
#endif
   use_comm = world > 1 || getenv ("NKP_FORCE_DIST") != NULL;       // the latter: exercise the RCCL path with one rank
   if (iam != 0 && world == 1) {
      // single-process run: extra ranks idle like ranks >= nprow*npcol do in the reference (:304)
      exit (EXIT_SUCCESS);
   }
   if (dbg_lvl) {
      printf ("(%d) dbg_lvl            = %d\n", iam, dbg_lvl);
      printf ("(%d) nprow              = %ld\n", iam, nprow);
      printf ("(%d) npcol              = %ld\n", iam, npcol);
      printf ("(%d) vars               = %s\n", iam, vars);
      printf ("(%d) matrix_fname       = %s\n", iam, matrix_fname);
      printf ("(%d) inout_fname        = %s\n\n", iam, inout_fname);
   }

#ifdef NKP_DIST
   // several ranks: every rank reads the row pointers and, once it knows its rows, only THEIR entries (hyperslab reads; the
   // reference has rank 0 read the whole matrix and send slices, src/solve_ABdist.c:141-225) -- at 0.25 degree the whole
   // matrix is 11 GB per rank
   const bool sliced = world > 1;
   if (sliced ? get_sparse_matrix_header (matrix_fname) : get_sparse_matrix (matrix_fname))
      exit (EXIT_FAILURE);
#else
   if (get_sparse_matrix (matrix_fname))
      exit (EXIT_FAILURE);
#endif
   if (dbg_lvl)
      printf ("(%d) row-oriented matrix read in\n", iam);
   // index maps before setup: the water-column boundaries come from them
   if (get_ind_maps (matrix_fname))
      exit (EXIT_FAILURE);
   if (coupled_tracer_cnt < 1 || (long long) coupled_tracer_cnt * tracer_state_len != flat_len) {
      fprintf (stderr, "(%d) coupled_tracer_cnt * tracer_state_len = %d * %d does not match flat_len = %d\n", iam, coupled_tracer_cnt,
               tracer_state_len, flat_len);
      exit (EXIT_FAILURE);
   }
   int nblk = 0;
   int_t *blk_start = nkp_column_blocks (&nblk);
   if (blk_start == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for blk_start\n", iam, argv[0]);
      exit (EXIT_FAILURE);
   }

   int *col_i = (int *) malloc ((size_t) (nblk ? nblk : 1) * sizeof (int));
   int *col_j = (int *) malloc ((size_t) (nblk ? nblk : 1) * sizeof (int));
   if (col_i == NULL || col_j == NULL || nkp_column_coords (nblk, col_i, col_j)) {
      fprintf (stderr, "(%d) could not derive the water-column grid positions in %s\n", iam, argv[0]);
      exit (EXIT_FAILURE);
   }

   nkp_options opt;
   nkp_default_options (&opt);
   opt.verbose = dbg_lvl;
   opt.rank = iam;
   opt.col_i = col_i;
   opt.col_j = col_j;
   options_from_env (&opt);

   // setup = the reference's factor-only call
   nkp_solver *solver = NULL;
   int fst_row = 0, m_loc = flat_len;
   printf ("(%d) calling %s\n", iam, prog_solver);
   fflush (stdout);
#ifdef NKP_DIST
   // the reference's partition rule (src/solve_ABdist.c:141-144), cuts snapped to water-column boundaries
   int fst_blk = 0, nblk_loc = nblk;
   nkp_comm_ops ops;
   memset (&ops, 0, sizeof ops);
   int info = 0;
   int file_transport = 0;
   // coupled tracers over several ranks are solved in cell-major order (NKP_CELL_MAJOR=0: the file's tracer-major rows)
   int32_t *cm_perm = NULL;                         // new row -> row of the file
   bool cell_major = use_comm && world > 1 && coupled_tracer_cnt > 1 && nblk % coupled_tracer_cnt == 0;
   { const char *e = getenv ("NKP_CELL_MAJOR"); if (e && atoi (e) == 0) cell_major = false; }
   if (use_comm) {
      // transport: RCCL (one GPU per rank; the unique id travels through NKP_RCCL_ID_FILE) or, for boxes with fewer GPUs
      // than ranks, the host-staged file transport (NKP_COMM=file, NKP_COMM_DIR=<directory every rank can write>)
      const char *transport = getenv ("NKP_COMM");
      file_transport = transport && strcmp (transport, "file") == 0;
      if (file_transport) {
         const char *dir = getenv ("NKP_COMM_DIR");
         int ndev = nkp_device_count ();
         if (dir == NULL || ndev < 1) {
            fprintf (stderr, "(%d) NKP_COMM=file needs NKP_COMM_DIR and a GPU\n", iam);
            exit (EXIT_FAILURE);
         }
         opt.device = local_rank % ndev;                 // ranks may share a device
         if (nkp_set_device (opt.device) || nkp_comm_file_init (&ops, dir, iam, world)) {
            fprintf (stderr, "(%d) file transport in %s could not be set up: %s\n", iam, dir, nkp_last_error ());
            exit (EXIT_FAILURE);
         }
      } else {
         // The id file carries a job tag (NKP_JOB_ID, else MASTER_PORT, else "-") in front of the 128-byte id: a rank only
         // accepts a file with ITS job's tag, so an id left behind by an earlier job on the same path is never taken, and
         // rank 0 removes the file as soon as the communicator exists (every rank has read it by then).
         struct { char tag[64]; unsigned char id[128]; } rec;
         memset (&rec, 0, sizeof rec);
         char tag[64];
         {
            const char *j = getenv ("NKP_JOB_ID");
            if (j == NULL || !*j) j = getenv ("MASTER_PORT");
            snprintf (tag, sizeof tag, "%s", (j && *j) ? j : "-");
         }
         const char *idfile = getenv ("NKP_RCCL_ID_FILE");
         if (idfile == NULL) {
            fprintf (stderr, "(%d) WORLD_SIZE = %d needs NKP_RCCL_ID_FILE (a path every rank can read)\n", iam, world);
            exit (EXIT_FAILURE);
         }
         opt.device = local_rank;
         if (nkp_set_device (local_rank)) {
            fprintf (stderr, "(%d) %s\n", iam, nkp_last_error ());
            exit (EXIT_FAILURE);
         }
         if (iam == 0) {
            char tmpname[4096];
            snprintf (tmpname, sizeof tmpname, "%s.tmp", idfile);
            snprintf (rec.tag, sizeof rec.tag, "%s", tag);
            FILE *f = NULL;
            if (nkp_comm_unique_id (rec.id) || (f = fopen (tmpname, "wb")) == NULL || fwrite (&rec, 1, sizeof rec, f) != sizeof rec || fclose (f) || rename (tmpname, idfile)) {
               fprintf (stderr, "(%d) could not publish the RCCL unique id through %s\n", iam, idfile);
               exit (EXIT_FAILURE);
            }
         } else {
            int ok = 0;
            const time_t launched = time (NULL);
            for (int tries = 0; tries < 1200 && !ok; tries++) {       // up to two minutes
               struct stat sb;
               // without a job tag fall back on the age test: only a file written around our launch can be ours
               if (stat (idfile, &sb) == 0 && (strcmp (tag, "-") != 0 || sb.st_mtime + 60 >= launched)) {
                  FILE *f = fopen (idfile, "rb");
                  if (f) {
                     ok = fread (&rec, 1, sizeof rec, f) == sizeof rec && strncmp (rec.tag, tag, sizeof rec.tag) == 0;
                     fclose (f);
                  }
               }
               if (!ok) usleep (100000);
            }
            if (!ok) {
               fprintf (stderr, "(%d) timed out waiting for the RCCL unique id of job '%s' in %s\n", iam, tag, idfile);
               exit (EXIT_FAILURE);
            }
         }
         if (nkp_comm_rccl_init (&ops, rec.id, iam, world)) {
            fprintf (stderr, "(%d) nkp_comm_rccl_init failed\n", iam);
            exit (EXIT_FAILURE);
         }
         if (iam == 0) (void) unlink (idfile);
      }
      if (!cell_major) nkp_rowblock_partition_snapped (blk_start, nblk, world, iam, &fst_row, &m_loc, &fst_blk, &nblk_loc);
   }
   if (cell_major) {
      // Coupled tracers over several ranks: the file's rows are tracer-major (src/matrix.c:778-784), so the reference's
      // contiguous row blocks would put the same-cell couplings (:955-961) off-rank in every row.  Renumber cell-major
      // (nkp_cell_major_order) and cut by the same rule between whole cells: every rank gets a latitude band of all tracers.
      const size_t nb = (size_t) nblk, nrow = (size_t) flat_len;
      cm_perm = (int32_t *) malloc ((nrow ? nrow : 1) * sizeof (int32_t));
      int32_t *inv = (int32_t *) malloc ((nrow ? nrow : 1) * sizeof (int32_t));
      int32_t *blk_new = (int32_t *) malloc ((nb + 1) * sizeof (int32_t));
      int32_t *col_t = (int32_t *) malloc ((nb ? nb : 1) * sizeof (int32_t));
      int32_t *col_src = (int32_t *) malloc ((nb ? nb : 1) * sizeof (int32_t));
      const int per = nblk / coupled_tracer_cnt;
      int_t *cell_start = (int_t *) malloc (((size_t) per + 1) * sizeof (int_t));
      if (!cm_perm || !inv || !blk_new || !col_t || !col_src || !cell_start) {
         fprintf (stderr, "(%d) malloc failed in %s for the cell-major ordering\n", iam, argv[0]);
         exit (EXIT_FAILURE);
      }
      if (nkp_cell_major_order (nblk, blk_start, coupled_tracer_cnt, cm_perm, blk_new, col_t, col_src)) {
         fprintf (stderr, "(%d) %s\n", iam, nkp_last_error ());
         exit (EXIT_FAILURE);
      }
      for (int r = 0; r < flat_len; r++) inv[cm_perm[r]] = r;
      for (int c = 0; c <= per; c++) cell_start[c] = blk_new[(size_t) c * coupled_tracer_cnt];
      int fst_cell = 0, ncell_loc = 0;
      nkp_rowblock_partition_snapped (cell_start, per, world, iam, &fst_row, &m_loc, &fst_cell, &ncell_loc);
      fst_blk = fst_cell * coupled_tracer_cnt;
      nblk_loc = ncell_loc * coupled_tracer_cnt;
      long long nnz_loc = 0;
      for (int r = fst_row; r < fst_row + m_loc; r++) nnz_loc += rowptr[cm_perm[r] + 1] - rowptr[cm_perm[r]];
      int_t *rowptr_loc = (int_t *) malloc ((size_t) (m_loc + 1) * sizeof (int_t));
      int_t *colind_loc = (int_t *) malloc ((size_t) (nnz_loc ? nnz_loc : 1) * sizeof (int_t));
      double *val_loc = (double *) malloc ((size_t) (nnz_loc ? nnz_loc : 1) * sizeof (double));
      int_t *blk_loc = (int_t *) malloc ((size_t) (nblk_loc + 1) * sizeof (int_t));
      int *ci_loc = (int *) malloc ((size_t) (nblk_loc ? nblk_loc : 1) * sizeof (int));
      int *cj_loc = (int *) malloc ((size_t) (nblk_loc ? nblk_loc : 1) * sizeof (int));
      if (!rowptr_loc || !colind_loc || !val_loc || !blk_loc || !ci_loc || !cj_loc) {
         fprintf (stderr, "(%d) malloc failed in %s for the local row block\n", iam, argv[0]);
         exit (EXIT_FAILURE);
      }
      if (sliced) {
         // the band's rows are `coupled_tracer_cnt` contiguous runs of the file (one per tracer): read those, then permute
         const int tsl = flat_len / coupled_tracer_cnt;
         const int lo0 = blk_start[fst_cell], hi0 = blk_start[fst_cell + ncell_loc];       // the band within tracer 0
         std::vector<size_t> ebase ((size_t) coupled_tracer_cnt + 1, 0);
         for (int t = 0; t < coupled_tracer_cnt; t++) ebase[(size_t) t + 1] = ebase[(size_t) t] + (size_t) (rowptr[t * tsl + hi0] - rowptr[t * tsl + lo0]);
         std::vector<int_t> col_c (ebase.back () + 1);
         std::vector<double> val_c (ebase.back () + 1);
         for (int t = 0; t < coupled_tracer_cnt; t++)
            if (get_sparse_matrix_rows (matrix_fname, t * tsl + lo0, t * tsl + hi0, col_c.data () + ebase[(size_t) t], val_c.data () + ebase[(size_t) t]))
               exit (EXIT_FAILURE);
         std::vector<std::pair<int_t, double>> buf;
         rowptr_loc[0] = 0;
         for (int r = 0; r < m_loc; r++) {
            const int old = cm_perm[fst_row + r], t = old / tsl;
            const size_t src = ebase[(size_t) t] + (size_t) (rowptr[old] - rowptr[t * tsl + lo0]);
            const int len = rowptr[old + 1] - rowptr[old];
            buf.clear ();
            for (int e = 0; e < len; e++) buf.push_back ({ inv[col_c[src + (size_t) e]], val_c[src + (size_t) e] });
            std::sort (buf.begin (), buf.end (), [] (const std::pair<int_t, double> &x, const std::pair<int_t, double> &y) { return x.first < y.first; });
            int_t o = rowptr_loc[r];
            for (const auto &pr : buf) { colind_loc[o] = pr.first; val_loc[o] = pr.second; o++; }
            rowptr_loc[r + 1] = o;
         }
      } else if (nkp_permuted_rows (flat_len, rowptr, colind, nzval_row_wise, cm_perm, inv, fst_row, fst_row + m_loc, rowptr_loc, colind_loc, val_loc)) {
         fprintf (stderr, "(%d) %s\n", iam, nkp_last_error ());
         exit (EXIT_FAILURE);
      }
      for (int b = 0; b <= nblk_loc; b++) blk_loc[b] = blk_new[fst_blk + b] - fst_row;
      for (int b = 0; b < nblk_loc; b++) { ci_loc[b] = col_i[col_src[fst_blk + b]]; cj_loc[b] = col_j[col_src[fst_blk + b]]; }
      opt.col_i = ci_loc;
      opt.col_j = cj_loc;
      opt.col_t = col_t + fst_blk;
      if (dbg_lvl)
         printf ("(%d) cell-major order: cells %d..%d of %d, fst_row, flat_len_loc, nnz_loc = %d, %d, %lld\n", iam, fst_cell, fst_cell + ncell_loc, per, fst_row, m_loc, nnz_loc);
      info = nkp_create_dist (&solver, &opt, flat_len, fst_row, m_loc, nnz_loc, rowptr_loc, colind_loc, val_loc, blk_loc, nblk_loc, coupled_tracer_cnt, &ops);
      free (rowptr_loc); free (colind_loc); free (val_loc); free (blk_loc); free (ci_loc); free (cj_loc);
      free (inv); free (blk_new); free (col_t); free (col_src); free (cell_start);
   } else {
      int_t *rowptr_loc = (int_t *) malloc ((size_t) (m_loc + 1) * sizeof (int_t));
      int_t *blk_loc = (int_t *) malloc ((size_t) (nblk_loc + 1) * sizeof (int_t));
      if (rowptr_loc == NULL || blk_loc == NULL) {
         fprintf (stderr, "(%d) malloc failed in %s for the local row block\n", iam, argv[0]);
         exit (EXIT_FAILURE);
      }
      const int_t e0 = rowptr[fst_row];
      for (int r = 0; r <= m_loc; r++) rowptr_loc[r] = rowptr[fst_row + r] - e0;          // rebased (:170-175)
      for (int b = 0; b <= nblk_loc; b++) blk_loc[b] = blk_start[fst_blk + b] - fst_row;
      opt.col_i = col_i + fst_blk;
      opt.col_j = col_j + fst_blk;
      if (dbg_lvl > 1)
         printf ("(%d) fst_row, flat_len_loc, nnz_loc = %d, %d, %d\n", iam, fst_row, m_loc, rowptr_loc[m_loc]);
      const int_t *colind_use = colind ? colind + e0 : NULL;
      const double *val_use = nzval_row_wise ? nzval_row_wise + e0 : NULL;
      std::vector<int_t> col_own;
      std::vector<double> val_own;
      if (sliced) {
         // this rank's colind / nzval slice, straight from the file (src/solve_ABdist.c:188-225 sends the same slices from rank 0)
         col_own.resize ((size_t) rowptr_loc[m_loc] + 1);
         val_own.resize ((size_t) rowptr_loc[m_loc] + 1);
         if (get_sparse_matrix_rows (matrix_fname, fst_row, fst_row + m_loc, col_own.data (), val_own.data ()))
            exit (EXIT_FAILURE);
         colind_use = col_own.data ();
         val_use = val_own.data ();
         if (dbg_lvl)
            printf ("(%d) read rows %d..%d of the matrix: %d of %d entries\n", iam, fst_row, fst_row + m_loc, rowptr_loc[m_loc], nnz);
      }
      info = nkp_create_dist (&solver, &opt, flat_len, fst_row, m_loc, rowptr_loc[m_loc], rowptr_loc, colind_use, val_use,
                              blk_loc, nblk_loc, coupled_tracer_cnt, use_comm ? &ops : NULL);
      free (rowptr_loc);
      free (blk_loc);
   }
#else
   int info = nkp_create (&solver, &opt, flat_len, nnz, rowptr, colind, nzval_row_wise, blk_start, nblk, coupled_tracer_cnt);
#endif
   if (dbg_lvl)
      printf ("(%d) %s info = %d\n", iam, prog_solver, info);
   if (info) {
      fprintf (stderr, "(%d) %s failed: %s\n", iam, prog_solver, nkp_last_error ());
      exit (EXIT_FAILURE);
   }
   free_sparse_matrix ();      // the device holds its own copy
   free (blk_start);
   free (col_i);
   free (col_j);

   char **vars_per_solve = (char **) malloc ((size_t) coupled_tracer_cnt * sizeof (char *));
   double *B = (double *) malloc ((size_t) (flat_len ? flat_len : 1) * sizeof (double));
   if (vars_per_solve == NULL || B == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for vars_per_solve / B\n", iam, argv[0]);
      exit (EXIT_FAILURE);
   }

   // NKP_RHS_CONCURRENCY=K (single-GPU flavour): up to K right-hand sides in flight at once, each on its own clone
   // of the solver (nkp_clone: own work vectors and stream, shared matrix and hierarchy) driven by its own host
   // thread.  Groups are still written back in the order given, up to the first failure, so the files end up
   // exactly as the sequential loop leaves them.
   int concurrency = 1;
   {
      const char *e = getenv ("NKP_RHS_CONCURRENCY");
      int v;
      if (e && !parse_to_int ((char *) e, &v) && v > 1 && !use_comm) concurrency = v;
   }
   // NKP_RHS_BLOCK=K (2..4, single-GPU flavour): K right-hand sides per nkp_solve call -- they share every sweep over the matrix
   // and the hierarchy (nkp.h, nkp_solve_batch_device), each with its own recurrence and stopping test, so every tracer
   // gets the bits of its own solve.  Written back in the order given, up to the first failure.
   int rhs_block = 1;
   {
      const char *e = getenv ("NKP_RHS_BLOCK");
      int v;
      if (e && !parse_to_int ((char *) e, &v) && v > 1 && !use_comm) rhs_block = v > 4 ? 4 : v;
   }
   if (rhs_block > 1) concurrency = 1;
   if (concurrency > 1 || rhs_block > 1) {
      std::vector<std::vector<char *>> groups;
      const char *sep = ",";
      for (char *var = strtok (vars, sep); var; var = strtok (NULL, sep)) {
         std::vector<char *> g;
         for (int t = 0; t < coupled_tracer_cnt; t++) {
            if (t > 0 && (var = strtok (NULL, sep)) == NULL) {
               fprintf (stderr, "(%d) error extracting tracer_ind=%d, ran out of var names\n", iam, t);
               exit (EXIT_FAILURE);
            }
            if (dbg_lvl)
               printf ("(%d) processing variable %s\n", iam, var);
            g.push_back (var);
         }
         groups.push_back (g);
      }
      const size_t ng = groups.size ();
      std::vector<std::vector<double>> Bs (ng);
      for (size_t g = 0; g < ng; g++) {
         Bs[g].resize ((size_t) (flat_len ? flat_len : 1));
         if (get_B_global (groups[g].data (), Bs[g].data ()))
            exit (EXIT_FAILURE);
      }
      struct result { int info = 0, iters = 0; double berr = 0.0, relres = 0.0; std::string err; };
      std::vector<result> res (ng);
      std::vector<nkp_solver *> handles (1, solver);
      if (rhs_block > 1) {
         printf ("(%d) calling nkp_solve for %zu right-hand sides, %d per call\n", iam, ng, rhs_block);
         fflush (stdout);
         const size_t ldb = (size_t) (flat_len ? flat_len : 1);
         std::vector<double> blockB ((size_t) rhs_block * ldb);
         for (size_t g0 = 0; g0 < ng; g0 += (size_t) rhs_block) {
            const int k = (int) (ng - g0 < (size_t) rhs_block ? ng - g0 : (size_t) rhs_block);
            double berr_k[4] = { 0, 0, 0, 0 }, relres_k[4] = { 0, 0, 0, 0 };
            int iters_k[4] = { 0, 0, 0, 0 };
            for (int c = 0; c < k; c++) memcpy (blockB.data () + (size_t) c * ldb, Bs[g0 + c].data (), ldb * sizeof (double));
            const int rc = nkp_solve (solver, blockB.data (), k, (int64_t) ldb, berr_k, iters_k, relres_k);
            const std::string err = rc ? nkp_last_error () : "";
            for (int c = 0; c < k; c++) {
               result &r = res[g0 + c];
               r.iters = iters_k[c]; r.berr = berr_k[c]; r.relres = relres_k[c];
               // the call returns the worst column's code; a block with a failed column counts as failed from its first tracer on
               // (nothing of the block is written: conservative next to the one-at-a-time loop, which would write the ones before it)
               r.info = rc;
               r.err = err;
               if (rc >= 0) memcpy (Bs[g0 + c].data (), blockB.data () + (size_t) c * ldb, ldb * sizeof (double));
            }
            if (rc < 0) break;
         }
      }
      while (rhs_block == 1 && handles.size () < (size_t) concurrency && handles.size () < ng) {
         nkp_solver *c = NULL;
         if (nkp_clone (solver, &c)) {            // out of device memory: run with what there is
            if (dbg_lvl)
               printf ("(%d) nkp_clone: %s; continuing with %zu right-hand sides in flight\n", iam, nkp_last_error (), handles.size ());
            break;
         }
         handles.push_back (c);
      }
      if (rhs_block == 1) {
         printf ("(%d) calling nkp_solve for %zu right-hand sides, %zu in flight\n", iam, ng, handles.size ());
         fflush (stdout);
      }
      std::vector<std::thread> workers;
      for (size_t w = 0; rhs_block == 1 && w < handles.size (); w++)
         workers.emplace_back ([&, w] () {
            for (size_t g = w; g < ng; g += handles.size ()) {
               result &r = res[g];
               r.info = nkp_solve (handles[w], Bs[g].data (), 1, flat_len, &r.berr, &r.iters, &r.relres);
               if (r.info) r.err = nkp_last_error ();       // thread-local: capture it on the thread that failed
            }
         });
      for (std::thread &t : workers) t.join ();
      for (size_t w = 1; w < handles.size (); w++) nkp_destroy (handles[w]);
      for (size_t g = 0; g < ng; g++) {
         result &r = res[g];
         if (dbg_lvl)
            printf ("(%d) nkp_solve info = %d, iterations = %d, relres = %.3e, berr = %.3e\n", iam, r.info, r.iters, r.relres, r.berr);
         r.info = berr_verdict (r.info, r.relres, r.berr);
         if (r.info) {
            fprintf (stderr, "(%d) nkp_solve failed (info = %d): %s\n(%d) %s left untouched in %s\n", iam, r.info, r.err.c_str (), iam,
                     groups[g][0], inout_fname);
            exit (EXIT_FAILURE);
         }
         if (put_B_global (groups[g].data (), Bs[g].data ()))
            exit (EXIT_FAILURE);
      }
   } else {
   // each group of coupled_tracer_cnt consecutive names is one right-hand side
      // (reference src/solve_ABglobal.c:370-409)
      const char *varsep = ",";
      for (char *var = strtok (vars, varsep); var; var = strtok (NULL, varsep)) {
         for (int t = 0; t < coupled_tracer_cnt; t++) {
            if (t > 0 && (var = strtok (NULL, varsep)) == NULL) {
               fprintf (stderr, "(%d) error extracting tracer_ind=%d, ran out of var names\n", iam, t);
               exit (EXIT_FAILURE);
            }
            if (dbg_lvl)
               printf ("(%d) processing variable %s\n", iam, var);
            if ((vars_per_solve[t] = strdup (var)) == NULL) {
               fprintf (stderr, "(%d) malloc failed in %s for vars_per_solve[%d]\n", iam, argv[0], t);
               exit (EXIT_FAILURE);
            }
         }
         if (get_B_global (vars_per_solve, B))
            exit (EXIT_FAILURE);
#ifdef NKP_DIST
         if (cm_perm) {                                  // the solver's rows are in cell-major order
            double *T = (double *) malloc ((size_t) (flat_len ? flat_len : 1) * sizeof (double));
            if (T == NULL) { fprintf (stderr, "(%d) malloc failed in %s for the reordered B\n", iam, argv[0]); exit (EXIT_FAILURE); }
            for (int r = 0; r < flat_len; r++) T[r] = B[cm_perm[r]];
            memcpy (B, T, (size_t) flat_len * sizeof (double));
            free (T);
         }
#endif

         double berr = 0.0, relres = 0.0;
         int iters = 0;
         printf ("(%d) calling nkp_solve\n", iam);
         fflush (stdout);
         // every rank flattened the whole B; it solves for its own slice (ldb = m_loc, src/solve_ABdist.c:571)
         info = nkp_solve (solver, B + fst_row, 1, m_loc, &berr, &iters, &relres);
         if (dbg_lvl)
            printf ("(%d) nkp_solve info = %d, iterations = %d, relres = %.3e, berr = %.3e\n", iam, info, iters, relres, berr);
         info = berr_verdict (info, relres, berr);
         if (info) {
            fprintf (stderr, "(%d) nkp_solve failed (info = %d): %s\n(%d) %s left untouched in %s\n", iam, info, nkp_last_error (), iam,
                     vars_per_solve[0], inout_fname);
            exit (EXIT_FAILURE);
         }
         if (use_comm) {
            // slices back to rank 0 (reference put_B_dist, src/solve_ABdist.c:377-406)
            double *X = (iam == 0) ? (double *) malloc ((size_t) (flat_len ? flat_len : 1) * sizeof (double)) : NULL;
            if (nkp_gather_root (solver, B + fst_row, X)) {
               fprintf (stderr, "(%d) %s\n", iam, nkp_last_error ());
               exit (EXIT_FAILURE);
            }
            if (iam == 0) {
#ifdef NKP_DIST
               if (cm_perm)
                  for (int r = 0; r < flat_len; r++) B[cm_perm[r]] = X[r];       // back to the file's tracer-major order
               else
#endif
               memcpy (B, X, (size_t) flat_len * sizeof (double));
               free (X);
            }
         }
         if (iam == 0 && put_B_global (vars_per_solve, B))
            exit (EXIT_FAILURE);
         for (int t = 0; t < coupled_tracer_cnt; t++)
            free (vars_per_solve[t]);
      }

   }

   nkp_destroy (solver);
#ifdef NKP_DIST
   if (use_comm) { if (file_transport) nkp_comm_file_free (&ops); else nkp_comm_rccl_free (&ops); }
#endif
   free_ind_maps ();
   free (vars_per_solve);
   free (vars);
   free (B);
   exit (EXIT_SUCCESS);
}
