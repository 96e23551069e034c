/* Matrix generator: index maps, sparsity pattern, coefficient assembly and the writers of
 * the matrix file (reference src/matrix.c:163-369 index maps, :466-981 pattern, :986-3617
 * terms, :3621-3840 clean-up + driver, :3844-3939 writer).
 *
 * The reference walks a coefficient cursor through the same chain of "is this neighbour
 * wet" tests in every one of its ~25 term routines.  Here that chain exists once
 * (row_layout): it yields, for a row, the slot of every possible stencil neighbour (or -1),
 * and the term routines address slots by name.  Rows only ever touch their own slots, so
 * every pass is a parallel loop over rows.  What is kept exactly: the pattern order inside
 * a row (it decides which duplicate survives sum_dup_vals), the order in which terms are
 * accumulated into a slot, and the form of every arithmetic expression -- so the values
 * are the ones the reference's code computes, to the last bit, given the same inputs.
 *
 * Units as in the reference: POP cgs, delta_t in seconds, sink rates per year.
 */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nc3_codec.h"
#include "nkp_host.h"

/* where the time goes, reported at dbg_lvl >= 1: field reads (I/O + type conversion) against row passes */
static double t_read = 0.0, t_rows = 0.0;

static double now_s (void)
{
   struct timespec ts;
   clock_gettime (CLOCK_MONOTONIC, &ts);
   return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

adv_opt_t adv_opt = adv_cent;
int l_adv_enforce_divfree = 1;
hmix_opt_t hmix_opt = hmix_isop_file;
vmix_opt_t vmix_opt = vmix_file;
char *tracer_fname = NULL;
per_tracer_opt_t *per_tracer_opt = NULL;
coupled_tracer_opt_t coupled_tracer_opt = coupled_tracer_none;

double delta_t;
double year_cnt;

static char *OCMIP_BGC_PO4_DOP_names[] = { "OCMIP_BGC_PO4", "OCMIP_BGC_DOP" };
static char *DIC_SHADOW_ALK_SHADOW_names[] = { "DIC_SHADOW", "ALK_SHADOW" };

int nkp_check_polar_rows (const char *subname);

static void trace (const char *what, const char *subname)
{
   if (dbg_lvl > 1) {
      printf ("(%d) %s %s\n", iam, what, subname);
      fflush (stdout);
   }
}

/* ------------------------------------------------------------------ index maps */
/* The flat state vector numbers the ocean cells with latitude rows outermost, then longitude, depth innermost, so
 * that every water column is one contiguous run (the ordering the reference fixes in src/matrix.c:239-251 and that
 * both solvers and the preconditioner's column blocks rely on).  Built by a prefix sum over the column depths:
 * column (j, i) owns the indices [first, first + KMT[j][i]). */
int gen_ind_maps (void)
{
   const char *who = "gen_ind_maps";
   long total = 0;

   trace ("entering", (char *) who);
   if (nkp_check_polar_rows ("comp_tracer_state_len")) {
      fprintf (stderr, "(%d) comp_tracer_state_len call failed in %s\n", iam, who);
      return 1;
   }
   for (int j = 0; j < jmt; j++)
      for (int i = 0; i < imt; i++) total += KMT[j][i];
   tracer_state_len = (int) total;
   if (dbg_lvl) printf ("(%d) tracer_state_len = %d\n\n", iam, tracer_state_len);

   int3_to_tracer_state_ind = malloc_3d_int (km, jmt, imt);
   tracer_state_ind_to_int3 = int3_to_tracer_state_ind ? (int3 *) malloc ((size_t) (total ? total : 1) * sizeof (int3)) : NULL;
   if (tracer_state_ind_to_int3 == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for %s\n", iam, who, int3_to_tracer_state_ind ? "tracer_state_ind_to_int3" : "int3_to_tracer_state_ind");
      return 1;
   }
   if (dbg_lvl > 2) printf ("(%d) mappings between flat and 3d indices\n", iam);
   int first = 0;
   for (int j = 0; j < jmt; j++)
      for (int i = 0; i < imt; i++) {
         const int depth = KMT[j][i];
         for (int k = 0; k < km; k++) int3_to_tracer_state_ind[k][j][i] = (k < depth) ? first + k : -1;
         for (int k = 0; k < depth; k++) {
            tracer_state_ind_to_int3[first + k] = (int3) { .i = i, .j = j, .k = k };
            if (dbg_lvl > 2) printf ("(%d) i = %3d, j = %3d, k = %2d, tracer_state_ind = %d\n", iam, i, j, k, first + k);
         }
         first += depth;
      }
   trace ("exiting", (char *) who);
   return 0;
}

/* the index-map section of the matrix file: the cube (with the -1 land marker declared as fill and missing value) and
 * the three inverse maps; schema as the reference writes it (src/matrix.c:263-369) */
int put_ind_maps (char *fname)
{
   char *who = "put_ind_maps";
   static const char *grid_dims[3] = { "z_t", "nlat", "nlon" };
   static const struct { const char *var; size_t member; } inverse[3] = {
      { "tracer_state_ind_to_i", offsetof (int3, i) }, { "tracer_state_ind_to_j", offsetof (int3, j) }, { "tracer_state_ind_to_k", offsetof (int3, k) } };
   static const char *marker_atts[2] = { "_FillValue", "missing_value" };
   const int land = -1;
   int status, cube_dims[3], len_dim, varid;
   nc3_file *f;

   trace ("entering", who);
   if ((status = nc3_open (fname, 1, &f))) return handle_nc_error (who, "nc_open", fname, status);
   if ((status = nc3_redef (f))) return handle_nc_error (who, "nc_redef", fname, status);
   if ((status = nc3_def_dim (f, "tracer_state_len", (size_t) tracer_state_len, &len_dim)))
      return handle_nc_error (who, "nc_def_dim", "tracer_state_len", status);
   for (int d = 0; d < 3; d++)
      if ((status = nc3_inq_dimid (f, grid_dims[d], &cube_dims[d]))) return handle_nc_error (who, "nc_inq_dimid", (char *) grid_dims[d], status);
   if ((status = nc3_def_var (f, "int3_to_tracer_state_ind", NC3_INT, 3, cube_dims, &varid)))
      return handle_nc_error (who, "nc_def_var", "int3_to_tracer_state_ind", status);
   if ((status = nc3_put_att_text (f, varid, "coordinates", strlen ("TLONG TLAT"), "TLONG TLAT")))
      return handle_nc_error (who, "nc_put_att_text", "int3_to_tracer_state_ind", status);
   for (int a = 0; a < 2; a++)
      if ((status = nc3_put_att_int (f, varid, marker_atts[a], NC3_INT, 1, &land)))
         return handle_nc_error (who, "nc_put_att_int", "int3_to_tracer_state_ind", status);
   for (int c = 0; c < 3; c++)
      if ((status = nc3_def_var (f, inverse[c].var, NC3_INT, 1, &len_dim, &varid)))
         return handle_nc_error (who, "nc_def_var", (char *) inverse[c].var, status);
   if ((status = nc3_close (f))) return handle_nc_error (who, "nc_close", fname, status);

   if (put_var_3d_int (fname, "int3_to_tracer_state_ind", int3_to_tracer_state_ind)) return 1;
   int *column = (int *) malloc ((size_t) (tracer_state_len ? tracer_state_len : 1) * sizeof (int));
   if (column == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for tracer_state_ind_to_ijk\n", iam, who);
      return 1;
   }
   int failed = 0;
   for (int c = 0; c < 3 && !failed; c++) {
      for (int s = 0; s < tracer_state_len; s++) column[s] = *(const int *) ((const char *) &tracer_state_ind_to_int3[s] + inverse[c].member);
      failed = put_var_1d_int (fname, (char *) inverse[c].var, column);
   }
   free (column);
   trace ("exiting", who);
   return failed ? 1 : 0;
}

/* ------------------------------------------------------------------ row layout */

enum {
   SELF, UP, DOWN, EAST, WEST, NORTH, SOUTH,                          /* 7-point neighbours   */
   UP2, DOWN2, EAST2, WEST2, NORTH2, SOUTH2,                          /* upwind3 second ring  */
   UP_EAST, DOWN_EAST, UP_WEST, DOWN_WEST, UP_NORTH, DOWN_NORTH, UP_SOUTH, DOWN_SOUTH,   /* isopycnal diagonals */
   NSLOT
};

typedef struct {
   int t, s;                  /* tracer, index inside the tracer's state vector */
   int i, j, k;
   int ip1, im1, ip2, im2;
   int kmt;                   /* KMT[j][i] */
   int slot[NSLOT];           /* offset inside the row, -1 = neighbour is not ocean */
   int vmix0;                 /* first of kmt vmix_matrix_file slots  */
   int sink0, sink_cnt;       /* generic-tracer sink slots: levels min(k,kmax) .. 0 */
   int other0;                /* coupled_tracer_cnt - 1 slots for the same cell of the other tracers */
   int len;
} row_ctx;

static int generic_sink_kmax (int t)
{
   int cnt = per_tracer_opt[t].sink_generic_tracer_depends_layer_cnt;
   return (cnt == -1) ? km - 1 : cnt - 1;
}

/* The one place that knows the pattern order (reference src/matrix.c:596-662 counts it,
 * :753-981 lays it out, every add_* routine re-walks it). */
static void row_layout (int t, int s, row_ctx *R)
{
   int i = tracer_state_ind_to_int3[s].i;
   int j = tracer_state_ind_to_int3[s].j;
   int k = tracer_state_ind_to_int3[s].k;
   int ip1 = (i < imt - 1) ? i + 1 : 0;
   int im1 = (i > 0) ? i - 1 : imt - 1;
   int ip2 = (ip1 < imt - 1) ? ip1 + 1 : 0;
   int im2 = (im1 > 0) ? im1 - 1 : imt - 1;
   int kmt = KMT[j][i];
   int n = 0;
   int *slot = R->slot;

   R->t = t; R->s = s;
   R->i = i; R->j = j; R->k = k;
   R->ip1 = ip1; R->im1 = im1; R->ip2 = ip2; R->im2 = im2;
   R->kmt = kmt;

#define PLACE(name, cond) slot[name] = (cond) ? n++ : -1
   PLACE (SELF, 1);
   PLACE (UP, k - 1 >= 0);
   PLACE (DOWN, k + 1 < kmt);
   PLACE (EAST, k < KMT[j][ip1]);
   PLACE (WEST, k < KMT[j][im1]);
   PLACE (NORTH, k < KMT[j + 1][i]);
   PLACE (SOUTH, k < KMT[j - 1][i]);
   if (adv_opt == adv_upwind3) {
      PLACE (UP2, k - 2 >= 0);
      PLACE (DOWN2, k + 2 < kmt);
      PLACE (EAST2, k < KMT[j][ip2]);
      PLACE (WEST2, k < KMT[j][im2]);
      PLACE (NORTH2, (j + 2 < jmt) && (k < KMT[j + 2][i]));
      PLACE (SOUTH2, (j - 2 >= 0) && (k < KMT[j - 2][i]));
   } else
      for (int c = UP2; c <= SOUTH2; c++) slot[c] = -1;
   if (hmix_opt == hmix_isop_file) {
      PLACE (UP_EAST, (k - 1 >= 0) && (k - 1 < KMT[j][ip1]));
      PLACE (DOWN_EAST, k + 1 < KMT[j][ip1]);
      PLACE (UP_WEST, (k - 1 >= 0) && (k - 1 < KMT[j][im1]));
      PLACE (DOWN_WEST, k + 1 < KMT[j][im1]);
      PLACE (UP_NORTH, (k - 1 >= 0) && (k - 1 < KMT[j + 1][i]));
      PLACE (DOWN_NORTH, k + 1 < KMT[j + 1][i]);
      PLACE (UP_SOUTH, (k - 1 >= 0) && (k - 1 < KMT[j - 1][i]));
      PLACE (DOWN_SOUTH, k + 1 < KMT[j - 1][i]);
   } else
      for (int c = UP_EAST; c <= DOWN_SOUTH; c++) slot[c] = -1;
#undef PLACE
   R->vmix0 = n;
   if (vmix_opt == vmix_matrix_file)
      n += kmt;
   R->sink0 = n;
   R->sink_cnt = 0;
   if (per_tracer_opt[t].sink_opt == sink_generic_tracer) {
      int kmax = generic_sink_kmax (t);
      R->sink_cnt = (k <= kmax) ? k + 1 : kmax + 1;
      n += R->sink_cnt;
   }
   R->other0 = n;
   n += coupled_tracer_cnt - 1;
   R->len = n;
}

/* column (inside one tracer's state vector) of every named slot */
static void row_columns (const row_ctx *R, int_t *col)
{
   static const int dk[NSLOT] = { 0, -1, 1, 0, 0, 0, 0, -2, 2, 0, 0, 0, 0, -1, 1, -1, 1, -1, 1, -1, 1 };
   static const int dj[NSLOT] = { 0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 2, -2, 0, 0, 0, 0, 1, 1, -1, -1 };
   int offset = R->t * tracer_state_len;
   int n;

   for (int c = 0; c < NSLOT; c++) {
      if (R->slot[c] < 0)
         continue;
      int ii = R->i;
      switch (c) {
      case EAST: case UP_EAST: case DOWN_EAST: ii = R->ip1; break;
      case WEST: case UP_WEST: case DOWN_WEST: ii = R->im1; break;
      case EAST2: ii = R->ip2; break;
      case WEST2: ii = R->im2; break;
      }
      col[R->slot[c]] = offset + int3_to_tracer_state_ind[R->k + dk[c]][R->j + dj[c]][ii];
   }
   n = R->vmix0;
   if (vmix_opt == vmix_matrix_file)
      for (int k2 = 0; k2 < R->kmt; k2++)
         col[n++] = offset + int3_to_tracer_state_ind[k2][R->j][R->i];
   n = R->sink0;
   for (int c = 0, k2 = R->sink_cnt - 1; c < R->sink_cnt; c++, k2--)
      col[n++] = offset + int3_to_tracer_state_ind[k2][R->j][R->i];
   n = R->other0;
   for (int t2 = 0; t2 < coupled_tracer_cnt; t2++)
      if (t2 != R->t)
         col[n++] = t2 * tracer_state_len + int3_to_tracer_state_ind[R->k][R->j][R->i];
}

/* ------------------------------------------------------------------ pattern */

static int init_matrix (void)
{
   char *subname = "init_matrix";
   long total = 0;

   trace ("entering", subname);
   flat_len = coupled_tracer_cnt * tracer_state_len;
   if (dbg_lvl)
      printf ("(%d) flat_len = %d\n\n", iam, flat_len);

   if ((rowptr = (int_t *) malloc ((size_t) (flat_len + 1) * sizeof (int_t))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for rowptr\n", iam, subname);
      return 1;
   }
   /* pass 1: row lengths -> rowptr (the reference's comp_nnz) */
   for (int t = 0; t < coupled_tracer_cnt; t++)
      for (int s = 0; s < tracer_state_len; s++) {
         row_ctx R;
         row_layout (t, s, &R);
         rowptr[t * tracer_state_len + s] = (int_t) total;
         total += R.len;
         if (total > 2147483647L) {
            fprintf (stderr, "(%d) %s: more than 2^31-1 matrix entries; the file schema is 32-bit\n", iam, subname);
            return 1;
         }
      }
   rowptr[flat_len] = (int_t) total;
   nnz = (int) total;
   if (dbg_lvl)
      printf ("(%d) nnz       = %d\n\n", iam, nnz);

   if ((nzval_row_wise = (double *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (double))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for nzval_row_wise\n", iam, subname);
      return 1;
   }
   if ((colind = (int_t *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (int_t))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for colind\n", iam, subname);
      return 1;
   }
   /* pass 2: zero values, column indices */
#pragma omp parallel for schedule(static)
   for (int row = 0; row < flat_len; row++) {
      row_ctx R;
      row_layout (row / tracer_state_len, row % tracer_state_len, &R);
      for (int c = 0; c < R.len; c++)
         nzval_row_wise[rowptr[row] + c] = 0.0;
      row_columns (&R, colind + rowptr[row]);
   }
   trace ("exiting", subname);
   return 0;
}

/* ------------------------------------------------------------------ field input */

/* the reference's "read, look the _FillValue up, zero it" sequence; want_fv = 0 for the
 * variables the reference reads without that treatment */
static double ***read_3d (char *fname, char *name, int want_fv)
{
   double ***F = malloc_3d_double (km, jmt, imt);
   double fv;
   double t0 = now_s ();

   if (F == NULL) {
      fprintf (stderr, "(%d) malloc failed for %s\n", iam, name);
      return NULL;
   }
   if (get_var_3d_double (fname, name, F)) {
      free_3d_double (F);
      return NULL;
   }
   if (want_fv) {
      size_t n = (size_t) km * (size_t) jmt * (size_t) imt;
      double *p = F[0][0];

      if (get_att_double (fname, name, "_FillValue", &fv)) {
         free_3d_double (F);
         return NULL;
      }
      for (size_t e = 0; e < n; e++)
         if (p[e] == fv)
            p[e] = 0.0;
   }
   t_read += now_s () - t0;
   return F;
}

static double **read_2d (char *fname, char *name, int want_fv)
{
   double **F = malloc_2d_double (jmt, imt);
   double fv;

   if (F == NULL) {
      fprintf (stderr, "(%d) malloc failed for %s\n", iam, name);
      return NULL;
   }
   if (get_var_2d_double (fname, name, F)) {
      free_2d_double (F);
      return NULL;
   }
   if (want_fv) {
      size_t n = (size_t) jmt * (size_t) imt;
      double *p = F[0];

      if (get_att_double (fname, name, "_FillValue", &fv)) {
         free_2d_double (F);
         return NULL;
      }
      for (size_t e = 0; e < n; e++)
         if (p[e] == fv)
            p[e] = 0.0;
   }
   return F;
}

static double ***zeros_3d (void)
{
   double ***F = malloc_3d_double (km, jmt, imt);
   if (F)
      memset (F[0][0], 0, (size_t) km * (size_t) jmt * (size_t) imt * sizeof (double));
   return F;
}

/* ------------------------------------------------------------------ row-parallel driver */

typedef void (*row_term) (const row_ctx *R, double *v, void *arg);

static void for_rows (row_term term, void *arg)
{
   double t0 = now_s ();
#pragma omp parallel for schedule(static)
   for (int row = 0; row < flat_len; row++) {
      row_ctx R;
      row_layout (row / tracer_state_len, row % tracer_state_len, &R);
      term (&R, nzval_row_wise + rowptr[row], arg);
   }
   t_rows += now_s () - t0;
}

static void for_rows_of_tracer (int t, row_term term, void *arg)
{
   double t0 = now_s ();
#pragma omp parallel for schedule(static)
   for (int s = 0; s < tracer_state_len; s++) {
      row_ctx R;
      row_layout (t, s, &R);
      term (&R, nzval_row_wise + rowptr[t * tracer_state_len + s], arg);
   }
   t_rows += now_s () - t0;
}

#define HAS(c) (R->slot[c] >= 0)
#define V(c) v[R->slot[c]]

/* ------------------------------------------------------------------ advection, centred / donor */

/* Volume transports through the faces of the T cells (what the reference assembles in src/matrix.c:986-1207).
 * A lateral face is bounded by two velocity points (B grid: velocities sit on the cell corners): its transport is the
 * mean of the two corner velocities times the corner spacing, each corner counting only where it is ocean (k < KMU);
 * with hmix_hor_file the eddy-induced (bolus) velocity of the face itself, times the face length, is added where the
 * cells on both sides are ocean.  One recipe per face orientation says which fields and which neighbours. */
typedef struct {
   const char *who;                            /* label of the -D1 progress lines */
   const char *vel, *corner_len;               /* corner velocity and the grid spacing that multiplies it */
   int tap_dj, tap_di;                         /* the face's second corner relative to corner (j, i) */
   const char *bolus, *face_len;               /* bolus velocity and face length (hmix_hor_file only) */
   int bolus_fill_aware;                       /* 0: the bolus field is read raw, fill values included */
   int across_dj, across_di;                   /* the T cell on the far side of the face */
} face_recipe;

static const face_recipe EAST_FACE = { "load_UTE", "UVEL", "DYU", -1, 0, "UISOP", "HTE", 0, 0, 1 };
static const face_recipe NORTH_FACE = { "load_VTN", "VVEL", "DXU", 0, -1, "VISOP", "HTN", 1, 1, 0 };

static double ***load_face_transport (const face_recipe *F)
{
   double ***T = zeros_3d ();
   double ***vel;
   double **len;

   if (T == NULL) return NULL;
   if (dbg_lvl) printf ("(%d) %s: reading %s,%s from %s\n", iam, F->who, F->vel, F->corner_len, circ_fname);
   if ((vel = read_3d (circ_fname, (char *) F->vel, 1)) == NULL || (len = read_2d (circ_fname, (char *) F->corner_len, 1)) == NULL) return NULL;
   for (int k = 0; k < km; k++)
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++) {
            const int cj[2] = { j, j + F->tap_dj }, ci[2] = { i, (i + F->tap_di + imt) % imt };
            for (int c = 0; c < 2; c++)
               if (k < KMU[cj[c]][ci[c]]) T[k][j][i] += 0.5 * vel[k][cj[c]][ci[c]] * len[cj[c]][ci[c]];
         }
   free_2d_double (len);
   free_3d_double (vel);
   if (hmix_opt != hmix_hor_file) return T;

   if (dbg_lvl) printf ("(%d) %s: reading %s,%s from %s\n", iam, F->who, F->bolus, F->face_len, circ_fname);
   if ((vel = read_3d (circ_fname, (char *) F->bolus, F->bolus_fill_aware)) == NULL || (len = read_2d (circ_fname, (char *) F->face_len, 1)) == NULL) return NULL;
   for (int k = 0; k < km; k++)
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++)
            if (k < KMT[j][i] && k < KMT[j + F->across_dj][(i + F->across_di) % imt]) T[k][j][i] += vel[k][j][i] * len[j][i];
   free_2d_double (len);
   free_3d_double (vel);
   return T;
}

static double ***load_UTE (void) { return load_face_transport (&EAST_FACE); }
static double ***load_VTN (void) { return load_face_transport (&NORTH_FACE); }

/* vertical velocity at the top of every ocean cell: resolved plus, with hmix_hor_file, bolus; rigid lid at the surface */
static double ***load_WVEL (void)
{
   static const char *parts[2] = { "WVEL", "WISOP" };
   const int nparts = (hmix_opt == hmix_hor_file) ? 2 : 1;
   double ***W = zeros_3d ();

   if (W == NULL) return NULL;
   for (int q = 0; q < nparts; q++) {
      double ***part;
      if (dbg_lvl) printf ("(%d) load_WVEL: reading %s from %s\n", iam, parts[q], circ_fname);
      if ((part = read_3d (circ_fname, (char *) parts[q], 1)) == NULL) return NULL;
      for (int k = 1; k < km; k++)
         for (int j = 1; j < jmt - 1; j++)
            for (int i = 0; i < imt; i++)
               if (k < KMT[j][i]) W[k][j][i] += part[k][j][i];
      free_3d_double (part);
   }
   return W;
}

/* weights of the cell's own value in the face interpolation: 1/0 donor, 0.5 centred
 * (reference src/matrix.c:1211-1451) */
static void term_UTE (const row_ctx *R, double *v, void *arg)
{
   double ***UTE = (double ***) arg;
   int i = R->i, j = R->j, k = R->k, im1 = R->im1;
   double east_self_interp_w = 0.5, west_self_interp_w = 0.5;

   if (adv_opt == adv_donor) {
      east_self_interp_w = (UTE[k][j][i] > 0.0) ? 1.0 : 0.0;
      west_self_interp_w = (UTE[k][j][im1] < 0.0) ? 1.0 : 0.0;
   }
   if (HAS (EAST))
      V (SELF) -= east_self_interp_w * UTE[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (WEST))
      V (SELF) += west_self_interp_w * UTE[k][j][im1] / TAREA[j][i] * delta_t;
   if (HAS (EAST))
      V (EAST) -= (1.0 - east_self_interp_w) * UTE[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (WEST))
      V (WEST) += (1.0 - west_self_interp_w) * UTE[k][j][im1] / TAREA[j][i] * delta_t;
}

static void term_VTN (const row_ctx *R, double *v, void *arg)
{
   double ***VTN = (double ***) arg;
   int i = R->i, j = R->j, k = R->k;
   double north_self_interp_w = 0.5, south_self_interp_w = 0.5;

   if (adv_opt == adv_donor) {
      north_self_interp_w = (VTN[k][j][i] > 0.0) ? 1.0 : 0.0;
      south_self_interp_w = (VTN[k][j - 1][i] < 0.0) ? 1.0 : 0.0;
   }
   if (HAS (NORTH))
      V (SELF) -= north_self_interp_w * VTN[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (SOUTH))
      V (SELF) += south_self_interp_w * VTN[k][j - 1][i] / TAREA[j][i] * delta_t;
   if (HAS (NORTH))
      V (NORTH) -= (1.0 - north_self_interp_w) * VTN[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (SOUTH))
      V (SOUTH) += (1.0 - south_self_interp_w) * VTN[k][j - 1][i] / TAREA[j][i] * delta_t;
}

static void term_WVEL (const row_ctx *R, double *v, void *arg)
{
   double ***WVEL = (double ***) arg;
   int i = R->i, j = R->j, k = R->k;
   double top_self_interp_w = 0.5, bot_self_interp_w = 0.5;

   if (adv_opt == adv_donor) {
      top_self_interp_w = (WVEL[k][j][i] > 0.0) ? 1.0 : 0.0;
      if (HAS (DOWN))
         bot_self_interp_w = (WVEL[k + 1][j][i] < 0.0) ? 1.0 : 0.0;
   }
   if (HAS (UP))
      V (SELF) -= top_self_interp_w * WVEL[k][j][i] / dz[k] * delta_t;
   if (HAS (DOWN))
      V (SELF) += bot_self_interp_w * WVEL[k + 1][j][i] / dz[k] * delta_t;
   if (HAS (UP))
      V (UP) -= (1.0 - top_self_interp_w) * WVEL[k][j][i] / dz[k] * delta_t;
   if (HAS (DOWN))
      V (DOWN) += (1.0 - bot_self_interp_w) * WVEL[k + 1][j][i] / dz[k] * delta_t;
}

/* ------------------------------------------------------------------ advection, upwind3 */

typedef struct {
   double ***POS, ***NEG;
   double *talfzp, *tbetzp, *tgamzp, *talfzm, *tbetzm, *tdelzm;
} upwind3_arg;

static int load_pos_neg (char *pos_name, char *neg_name, int zero_surface, upwind3_arg *U)
{
   if (dbg_lvl)
      printf ("(%d) %s: reading %s,%s from %s\n", iam, "load_upwind3", pos_name, neg_name, circ_fname);
   if ((U->POS = read_3d (circ_fname, pos_name, 1)) == NULL)
      return 1;
   if ((U->NEG = read_3d (circ_fname, neg_name, 1)) == NULL)
      return 1;
   if (zero_surface)
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++) {
            U->POS[0][j][i] = 0.0;
            U->NEG[0][j][i] = 0.0;
         }
   return 0;
}

/* quadratic upstream interpolation on a uniform horizontal index space: face value =
 * 3/8 downstream + 3/4 upstream - 1/8 far-upstream; a far-upstream cell that is land
 * gives its weight to the upstream cell (reference src/matrix.c:1578-1817) */
static void term_UTE_upwind3 (const row_ctx *R, double *v, void *arg)
{
   upwind3_arg *U = (upwind3_arg *) arg;
   double ***UTE_POS = U->POS, ***UTE_NEG = U->NEG;
   int i = R->i, j = R->j, k = R->k, im1 = R->im1;

   /* east face */
   if (HAS (WEST))
      V (SELF) -= 0.75 * UTE_POS[k][j][i] / TAREA[j][i] * delta_t;
   else
      V (SELF) -= (0.75 - 0.125) * UTE_POS[k][j][i] / TAREA[j][i] * delta_t;
   V (SELF) -= 0.375 * UTE_NEG[k][j][i] / TAREA[j][i] * delta_t;
   /* west face */
   V (SELF) += 0.375 * UTE_POS[k][j][im1] / TAREA[j][i] * delta_t;
   if (HAS (EAST))
      V (SELF) += 0.75 * UTE_NEG[k][j][im1] / TAREA[j][i] * delta_t;
   else
      V (SELF) += (0.75 - 0.125) * UTE_NEG[k][j][im1] / TAREA[j][i] * delta_t;

   if (HAS (EAST)) {
      V (EAST) -= 0.375 * UTE_POS[k][j][i] / TAREA[j][i] * delta_t;
      if (HAS (EAST2))
         V (EAST) -= 0.75 * UTE_NEG[k][j][i] / TAREA[j][i] * delta_t;
      else
         V (EAST) -= (0.75 - 0.125) * UTE_NEG[k][j][i] / TAREA[j][i] * delta_t;
      V (EAST) += (-0.125) * UTE_NEG[k][j][im1] / TAREA[j][i] * delta_t;
   }
   if (HAS (WEST)) {
      V (WEST) -= (-0.125) * UTE_POS[k][j][i] / TAREA[j][i] * delta_t;
      if (HAS (WEST2))
         V (WEST) += 0.75 * UTE_POS[k][j][im1] / TAREA[j][i] * delta_t;
      else
         V (WEST) += (0.75 - 0.125) * UTE_POS[k][j][im1] / TAREA[j][i] * delta_t;
      V (WEST) += 0.375 * UTE_NEG[k][j][im1] / TAREA[j][i] * delta_t;
   }
   if (HAS (EAST2))
      V (EAST2) -= (-0.125) * UTE_NEG[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (WEST2))
      V (WEST2) += (-0.125) * UTE_POS[k][j][im1] / TAREA[j][i] * delta_t;
}

static void term_VTN_upwind3 (const row_ctx *R, double *v, void *arg)
{
   upwind3_arg *U = (upwind3_arg *) arg;
   double ***VTN_POS = U->POS, ***VTN_NEG = U->NEG;
   int i = R->i, j = R->j, k = R->k;

   /* north face */
   if (HAS (SOUTH))
      V (SELF) -= 0.75 * VTN_POS[k][j][i] / TAREA[j][i] * delta_t;
   else
      V (SELF) -= (0.75 - 0.125) * VTN_POS[k][j][i] / TAREA[j][i] * delta_t;
   V (SELF) -= 0.375 * VTN_NEG[k][j][i] / TAREA[j][i] * delta_t;
   /* south face */
   V (SELF) += 0.375 * VTN_POS[k][j - 1][i] / TAREA[j][i] * delta_t;
   if (HAS (NORTH))
      V (SELF) += 0.75 * VTN_NEG[k][j - 1][i] / TAREA[j][i] * delta_t;
   else
      V (SELF) += (0.75 - 0.125) * VTN_NEG[k][j - 1][i] / TAREA[j][i] * delta_t;

   if (HAS (NORTH)) {
      V (NORTH) -= 0.375 * VTN_POS[k][j][i] / TAREA[j][i] * delta_t;
      if (HAS (NORTH2))
         V (NORTH) -= 0.75 * VTN_NEG[k][j][i] / TAREA[j][i] * delta_t;
      else
         V (NORTH) -= (0.75 - 0.125) * VTN_NEG[k][j][i] / TAREA[j][i] * delta_t;
      V (NORTH) += (-0.125) * VTN_NEG[k][j - 1][i] / TAREA[j][i] * delta_t;
   }
   if (HAS (SOUTH)) {
      V (SOUTH) -= (-0.125) * VTN_POS[k][j][i] / TAREA[j][i] * delta_t;
      if (HAS (SOUTH2))
         V (SOUTH) += 0.75 * VTN_POS[k][j - 1][i] / TAREA[j][i] * delta_t;
      else
         V (SOUTH) += (0.75 - 0.125) * VTN_POS[k][j - 1][i] / TAREA[j][i] * delta_t;
      V (SOUTH) += 0.375 * VTN_NEG[k][j - 1][i] / TAREA[j][i] * delta_t;
   }
   if (HAS (NORTH2))
      V (NORTH2) -= (-0.125) * VTN_NEG[k][j][i] / TAREA[j][i] * delta_t;
   if (HAS (SOUTH2))
      V (SOUTH2) += (-0.125) * VTN_POS[k][j - 1][i] / TAREA[j][i] * delta_t;
}

/* vertical: the same three-point upstream interpolation with weights for the stretched
 * grid (POP's talfzp .. tdelzm, reference src/matrix.c:1868-1903) */
static int upwind3_vertical_weights (upwind3_arg *U)
{
   char *subname = "add_WVEL_coeffs_upwind3";
   double *w = (double *) calloc ((size_t) (7 * km + 2), sizeof (double));
   double *dzc;

   if (w == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for dzc\n", iam, subname);
      return 1;
   }
   U->talfzp = w; U->tbetzp = w + km; U->tgamzp = w + 2 * km;
   U->talfzm = w + 3 * km; U->tbetzm = w + 4 * km; U->tdelzm = w + 5 * km;
   dzc = w + 6 * km + 1;                     /* dzc[-1 .. km] */
   dzc[-1] = dz[0];
   for (int k = 0; k < km; k++)
      dzc[k] = dz[k];
   dzc[km] = dzc[km - 1];
   for (int k = 0; k < km - 1; k++) {
      U->talfzp[k] = dz[k] * (2.0 * dz[k] + dzc[k - 1]) / (dz[k] + dz[k + 1]) / (dzc[k - 1] + 2.0 * dz[k] + dz[k + 1]);
      U->tbetzp[k] = dz[k + 1] * (2.0 * dz[k] + dzc[k - 1]) / (dz[k] + dz[k + 1]) / (dz[k] + dzc[k - 1]);
      U->tgamzp[k] = -(dz[k] * dz[k + 1]) / (dz[k] + dzc[k - 1]) / (dz[k + 1] + dzc[k - 1] + 2.0 * dz[k]);
   }
   U->tbetzp[0] = U->tbetzp[0] + U->tgamzp[0];
   U->tgamzp[0] = 0.0;
   U->talfzp[km - 1] = 0.0;
   U->tbetzp[km - 1] = 0.0;
   U->tgamzp[km - 1] = 0.0;
   for (int k = 0; k < km - 1; k++) {
      U->talfzm[k] = dz[k] * (2.0 * dz[k + 1] + dzc[k + 2]) / (dz[k] + dz[k + 1]) / (dz[k + 1] + dzc[k + 2]);
      U->tbetzm[k] = dz[k + 1] * (2.0 * dz[k + 1] + dzc[k + 2]) / (dz[k] + dz[k + 1]) / (dz[k] + dzc[k + 2] + 2.0 * dz[k + 1]);
      U->tdelzm[k] = -(dz[k] * dz[k + 1]) / (dz[k + 1] + dzc[k + 2]) / (dz[k] + dzc[k + 2] + 2.0 * dz[k + 1]);
   }
   U->talfzm[km - 1] = 0.0;
   U->tbetzm[km - 1] = 0.0;
   U->tdelzm[km - 1] = 0.0;
   return 0;
}

static void term_WVEL_upwind3 (const row_ctx *R, double *v, void *arg)
{
   upwind3_arg *U = (upwind3_arg *) arg;
   double ***WVEL_POS = U->POS, ***WVEL_NEG = U->NEG;
   const double *talfzp = U->talfzp, *tbetzp = U->tbetzp, *tgamzp = U->tgamzp;
   const double *talfzm = U->talfzm, *tbetzm = U->tbetzm, *tdelzm = U->tdelzm;
   int i = R->i, j = R->j, k = R->k;

   if (HAS (UP)) {                     /* top face */
      if (HAS (DOWN))
         V (SELF) -= talfzm[k - 1] * WVEL_POS[k][j][i] / dz[k] * delta_t;
      else
         V (SELF) -= (talfzm[k - 1] + tdelzm[k - 1]) * WVEL_POS[k][j][i] / dz[k] * delta_t;
      V (SELF) -= talfzp[k - 1] * WVEL_NEG[k][j][i] / dz[k] * delta_t;
   }
   if (HAS (DOWN)) {                   /* bottom face */
      V (SELF) += tbetzm[k] * WVEL_POS[k + 1][j][i] / dz[k] * delta_t;
      V (SELF) += tbetzp[k] * WVEL_NEG[k + 1][j][i] / dz[k] * delta_t;
   }
   if (HAS (UP)) {
      V (UP) -= tbetzm[k - 1] * WVEL_POS[k][j][i] / dz[k] * delta_t;
      V (UP) -= tbetzp[k - 1] * WVEL_NEG[k][j][i] / dz[k] * delta_t;
      if (HAS (DOWN))
         V (UP) += tgamzp[k] * WVEL_NEG[k + 1][j][i] / dz[k] * delta_t;
   }
   if (HAS (DOWN)) {
      if (HAS (UP))
         V (DOWN) -= tdelzm[k - 1] * WVEL_POS[k][j][i] / dz[k] * delta_t;
      if (HAS (DOWN2))
         V (DOWN) += talfzm[k] * WVEL_POS[k + 1][j][i] / dz[k] * delta_t;
      else
         V (DOWN) += (talfzm[k] + tdelzm[k]) * WVEL_POS[k + 1][j][i] / dz[k] * delta_t;
      V (DOWN) += talfzp[k] * WVEL_NEG[k + 1][j][i] / dz[k] * delta_t;
   }
   if (HAS (UP2))
      V (UP2) -= tgamzp[k - 1] * WVEL_NEG[k][j][i] / dz[k] * delta_t;
   if (HAS (DOWN2))
      V (DOWN2) += tdelzm[k] * WVEL_POS[k + 1][j][i] / dz[k] * delta_t;
}

static int add_adv (void)
{
   char *subname = "add_adv";
   double ***VEL_WIDTH;
   upwind3_arg U;

   trace ("entering", subname);
   switch (adv_opt) {
   case adv_none:
      break;
   case adv_donor:
   case adv_cent:
      if ((VEL_WIDTH = load_UTE ()) == NULL)
         return 1;
      for_rows (term_UTE, VEL_WIDTH);
      free_3d_double (VEL_WIDTH);
      if ((VEL_WIDTH = load_VTN ()) == NULL)
         return 1;
      for_rows (term_VTN, VEL_WIDTH);
      free_3d_double (VEL_WIDTH);
      if ((VEL_WIDTH = load_WVEL ()) == NULL)
         return 1;
      for_rows (term_WVEL, VEL_WIDTH);
      free_3d_double (VEL_WIDTH);
      break;
   case adv_upwind3:
      memset (&U, 0, sizeof (U));
      if (load_pos_neg ("UTE_POS", "UTE_NEG", 0, &U))
         return 1;
      for_rows (term_UTE_upwind3, &U);
      free_3d_double (U.POS);
      free_3d_double (U.NEG);
      if (load_pos_neg ("VTN_POS", "VTN_NEG", 0, &U))
         return 1;
      for_rows (term_VTN_upwind3, &U);
      free_3d_double (U.POS);
      free_3d_double (U.NEG);
      if (load_pos_neg ("WTK_POS", "WTK_NEG", 1, &U))
         return 1;
      if (upwind3_vertical_weights (&U))
         return 1;
      for_rows (term_WVEL_upwind3, &U);
      free (U.talfzp);
      free_3d_double (U.POS);
      free_3d_double (U.NEG);
      break;
   }
   if (dbg_lvl)
      printf ("(%d) adv terms added\n\n", iam);
   trace ("exiting", subname);
   return 0;
}

/* Replace the diagonal by minus the sum of the other advective weights, which removes the
 * residual divergence of the input velocities from the operator (constants are then exactly
 * in the null space of the advection term; reference src/matrix.c:2094-2207).  Must run
 * right after add_adv: it sums whatever is in the slots at that point. */
static void term_divfree (const row_ctx *R, double *v, void *arg)
{
   double nzval_sum_non_self = 0.0;

   (void) arg;
   for (int c = UP; c <= SOUTH2; c++)
      if (HAS (c))
         nzval_sum_non_self += V (c);
   V (SELF) = -nzval_sum_non_self;
}

/* ------------------------------------------------------------------ lateral mixing */

typedef struct { double **HUS, **HTE, **HUW, **HTN; double ***KAPPA; double ah; } hmix_arg;

static int load_hmix_metrics (char *subname, hmix_arg *H)
{
   if (dbg_lvl)
      printf ("(%d) %s: reading HUS,HTE,HUW,HTN from %s\n", iam, subname, circ_fname);
   if ((H->HUS = read_2d (circ_fname, "HUS", 1)) == NULL) return 1;
   if ((H->HTE = read_2d (circ_fname, "HTE", 1)) == NULL) return 1;
   if ((H->HUW = read_2d (circ_fname, "HUW", 1)) == NULL) return 1;
   if ((H->HTN = read_2d (circ_fname, "HTN", 1)) == NULL) return 1;
   return 0;
}

static void free_hmix_metrics (hmix_arg *H)
{
   free_2d_double (H->HTN);
   free_2d_double (H->HUW);
   free_2d_double (H->HTE);
   free_2d_double (H->HUS);
}

/* Laplacian with face diffusivity kappa: flux coefficient = kappa * face length / centre
 * distance / cell area (reference src/matrix.c:2391-2726; const uses ah = 4.0e6 cm^2/s) */
static void term_hmix_laplacian (const row_ctx *R, double *v, void *arg)
{
   hmix_arg *H = (hmix_arg *) arg;
   double **HUS = H->HUS, **HTE = H->HTE, **HUW = H->HUW, **HTN = H->HTN;
   double ***KAPPA = H->KAPPA;
   double ah = H->ah;
   int i = R->i, j = R->j, k = R->k, ip1 = R->ip1, im1 = R->im1;
   double ce = 0.0, cw = 0.0, cn = 0.0, cs = 0.0;

   if (KAPPA == NULL) {
      if (HAS (EAST)) ce = ah * HTE[j][i] / HUS[j][i] / TAREA[j][i] * delta_t;
      if (HAS (WEST)) cw = ah * HTE[j][im1] / HUS[j][im1] / TAREA[j][i] * delta_t;
      if (HAS (NORTH)) cn = ah * HTN[j][i] / HUW[j][i] / TAREA[j][i] * delta_t;
      if (HAS (SOUTH)) cs = ah * HTN[j - 1][i] / HUW[j - 1][i] / TAREA[j][i] * delta_t;
   } else {
      if (HAS (EAST)) ce = 0.5 * (KAPPA[k][j][i] + KAPPA[k][j][ip1]) * HTE[j][i] / HUS[j][i] / TAREA[j][i] * delta_t;
      if (HAS (WEST)) cw = 0.5 * (KAPPA[k][j][im1] + KAPPA[k][j][i]) * HTE[j][im1] / HUS[j][im1] / TAREA[j][i] * delta_t;
      if (HAS (NORTH)) cn = 0.5 * (KAPPA[k][j][i] + KAPPA[k][j + 1][i]) * HTN[j][i] / HUW[j][i] / TAREA[j][i] * delta_t;
      if (HAS (SOUTH)) cs = 0.5 * (KAPPA[k][j - 1][i] + KAPPA[k][j][i]) * HTN[j - 1][i] / HUW[j - 1][i] / TAREA[j][i] * delta_t;
   }
   V (SELF) -= (ce + cw + cn + cs);
   if (HAS (EAST)) V (EAST) += ce;
   if (HAS (WEST)) V (WEST) += cw;
   if (HAS (NORTH)) V (NORTH) += cn;
   if (HAS (SOUTH)) V (SOUTH) += cs;
}

static int add_hmix_const (void)
{
   char *subname = "add_hmix_const";
   hmix_arg H;

   trace ("entering", subname);
   memset (&H, 0, sizeof (H));
   H.ah = 4.0e6;
   if (load_hmix_metrics (subname, &H))
      return 1;
   for_rows (term_hmix_laplacian, &H);
   free_hmix_metrics (&H);
   trace ("exiting", subname);
   return 0;
}

static int add_hmix_hor_file (void)
{
   char *subname = "add_hmix_hor_file";
   hmix_arg H;
   double ***WORK;

   trace ("entering", subname);
   memset (&H, 0, sizeof (H));
   if (dbg_lvl)
      printf ("(%d) %s: reading KAPPA_ISOP,HOR_DIFF from %s\n", iam, subname, circ_fname);
   if ((H.KAPPA = read_3d (circ_fname, "KAPPA_ISOP", 1)) == NULL)
      return 1;
   if ((WORK = read_3d (circ_fname, "HOR_DIFF", 1)) == NULL)
      return 1;
   for (int k = 0; k < km; k++)
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++)
            if (k < KMT[j][i])
               H.KAPPA[k][j][i] += WORK[k][j][i];
   free_3d_double (WORK);
   if (load_hmix_metrics (subname, &H))
      return 1;
   for_rows (term_hmix_laplacian, &H);
   free_hmix_metrics (&H);
   free_3d_double (H.KAPPA);
   trace ("exiting", subname);
   return 0;
}

/* Isopycnal mixing from impulse-response fields.  The model was run with unit impulses on a
 * 4 x 3 x 3 colouring of the grid, so the field coloured (i'%4, j'%3, k'%3), sampled at cell
 * (i,j,k), is the tendency of (i,j,k) per unit tracer at its stencil neighbour (i',j',k') --
 * i.e. the matrix entry for that neighbour (reference src/matrix.c:2211-2387). */
typedef struct { double ***IRF; int iprime, jprime, kprime; } irf_arg;

static void term_hmix_isop (const row_ctx *R, double *v, void *arg)
{
   static const int dk[NSLOT] = { 0, -1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 1, -1, 1, -1, 1, -1, 1 };
   static const int dj[NSLOT] = { 0, 0, 0, 0, 0, 1, -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, -1, -1 };
   irf_arg *A = (irf_arg *) arg;

   for (int c = SELF; c <= DOWN_SOUTH; c++) {
      if (c >= UP2 && c <= SOUTH2)
         continue;
      if (!HAS (c))
         continue;
      int ii = R->i;
      switch (c) {
      case EAST: case UP_EAST: case DOWN_EAST: ii = R->ip1; break;
      case WEST: case UP_WEST: case DOWN_WEST: ii = R->im1; break;
      }
      if ((ii % 4 == A->iprime) && ((R->j + dj[c]) % 3 == A->jprime) && ((R->k + dk[c]) % 3 == A->kprime))
         V (c) += A->IRF[R->k][R->j][R->i] * delta_t;
   }
}

/* The 36 impulse-response fields of the isopycnal mixing operator (reference src/matrix.c:2233-2259): field q of the
 * (4, 3, 3) colouring is stored under one of the spellings below, tried in this order. */
static const char *const irf_spellings[] = { "HDIF_EXPLICIT_3D_IRF_%d_%d_%d", "HDIF_EXPLICIT_3D_IRF_NK_%d_%d_%d" };

/* name of the colouring's field (ip, jp, kp) as the circulation file spells it; 0 found, 1 absent under every spelling, -1 I/O error */
static int irf_field_name (const char *subname, int ip, int jp, int kp, char *name, size_t cap)
{
   for (size_t v = 0; v < sizeof irf_spellings / sizeof irf_spellings[0]; v++) {
      int present = 0;
      snprintf (name, cap, irf_spellings[v], ip + 1, jp + 1, kp + 1);
      if (var_exists_in_file (circ_fname, name, &present)) {
         fprintf (stderr, "(%d) var_exists_in_file failed in %s for field_name %s in file %s\n", iam, subname, name, circ_fname);
         return -1;
      }
      if (present)
         return 0;
      if (dbg_lvl)
         printf ("(%d) %s: %s not found in %s\n", iam, subname, name, circ_fname);
   }
   return 1;
}

static int add_hmix_isop_file (void)
{
   char *subname = "add_hmix_isop_file";
   irf_arg A;
   char field[64];
   int rc = 0;

   trace ("entering", subname);
   if ((A.IRF = malloc_3d_double (km, jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for IRF\n", iam, subname);
      return 1;
   }
   /* q enumerates the colouring with kprime fastest, like the reference's loop nest */
   for (int q = 0; q < 4 * 3 * 3 && !rc; q++) {
      A.iprime = q / 9;
      A.jprime = (q / 3) % 3;
      A.kprime = q % 3;
      if (irf_field_name (subname, A.iprime, A.jprime, A.kprime, field, sizeof field)) {
         rc = 1;
         break;
      }
      if (dbg_lvl)
         printf ("(%d) %s: reading %s from %s\n", iam, subname, field, circ_fname);
      const double t0 = now_s ();
      rc = get_var_3d_double (circ_fname, field, A.IRF) != 0;
      t_read += now_s () - t0;
      if (!rc)
         for_rows (term_hmix_isop, &A);
   }
   if (rc)
      return 1;
   free_3d_double (A.IRF);
   trace ("exiting", subname);
   return 0;
}

/* lateral mixing by option: the routine that adds it, and the advection scheme it cannot be combined with (-1: none) */
static const struct {
   int (*add) (void);
   int refuses_adv;
   const char *why;
} hmix_table[] = {
   [hmix_none] = { NULL, -1, NULL },
   [hmix_const] = { add_hmix_const, -1, NULL },
   [hmix_hor_file] = { add_hmix_hor_file, adv_upwind3, "cannot use hmix_hor_file with adv_upwind3" },
   [hmix_isop_file] = { add_hmix_isop_file, -1, NULL },
};

static int add_hmix (void)
{
   char *subname = "add_hmix";

   trace ("entering", subname);
   if ((size_t) hmix_opt < sizeof hmix_table / sizeof hmix_table[0] && hmix_table[hmix_opt].add) {
      if (hmix_table[hmix_opt].refuses_adv == (int) adv_opt) {
         fprintf (stderr, "(%d) %s\n", iam, hmix_table[hmix_opt].why);
         return 1;
      }
      if (hmix_table[hmix_opt].add ())
         return 1;
   }
   if (dbg_lvl)
      printf ("(%d) hmix terms added\n\n", iam);
   trace ("exiting", subname);
   return 0;
}

/* ------------------------------------------------------------------ vertical mixing */

/* diffusivity at the interface below level k sits at index k; interface distance is the
 * mean of the two layer thicknesses (reference src/matrix.c:2842-3014; const: 0.1 cm^2/s) */
static void term_vmix (const row_ctx *R, double *v, void *arg)
{
   double ***VDC = (double ***) arg;
   int i = R->i, j = R->j, k = R->k;
   double vdc = 0.1;
   double ct = 0.0, cb = 0.0;

   if (HAS (UP))
      ct = (VDC ? VDC[k - 1][j][i] : vdc) / (0.5 * (dz[k - 1] + dz[k])) / dz[k] * delta_t;
   if (HAS (DOWN))
      cb = (VDC ? VDC[k][j][i] : vdc) / (0.5 * (dz[k] + dz[k + 1])) / dz[k] * delta_t;
   V (SELF) -= (ct + cb);
   if (HAS (UP)) V (UP) += ct;
   if (HAS (DOWN)) V (DOWN) += cb;
}

static int add_vmix_file (void)
{
   char *subname = "add_vmix_file";
   double ***VDC_TOTAL, ***VDC_READ;

   trace ("entering", subname);
   if (dbg_lvl)
      printf ("(%d) %s: reading %s from %s for VDC\n", iam, subname, "VDC_S", circ_fname);
   if ((VDC_TOTAL = read_3d (circ_fname, "VDC_S", 1)) == NULL)
      return 1;
   if (dbg_lvl)
      printf ("(%d) %s: reading %s from %s for VDC\n", iam, subname, "VDC_GM", circ_fname);
   if ((VDC_READ = read_3d (circ_fname, "VDC_GM", 1)) == NULL)
      return 1;
   for (int k = 0; k < km; k++)
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++)
            VDC_TOTAL[k][j][i] += VDC_READ[k][j][i];
   free_3d_double (VDC_READ);
   for_rows (term_vmix, VDC_TOTAL);
   free_3d_double (VDC_TOTAL);
   trace ("exiting", subname);
   return 0;
}

/* whole-column implicit-mixing operator: variable vmix_matrix_%03d_CUR at (k,j,i) is the
 * entry (row k, column level - 1) of that water column's matrix (reference :2776-2838) */
typedef struct { double ***F; int kprime; } vmix_matrix_arg;

static void term_vmix_matrix (const row_ctx *R, double *v, void *arg)
{
   vmix_matrix_arg *A = (vmix_matrix_arg *) arg;

   if (A->kprime < R->kmt)
      v[R->vmix0 + A->kprime] += A->F[R->k][R->j][R->i] * delta_t;
}

static int add_vmix_matrix_file (void)
{
   char *subname = "add_vmix_matrix_file";
   vmix_matrix_arg A;
   char varname[64];

   trace ("entering", subname);
   if ((A.F = malloc_3d_double (km, jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for vmix_matrix_var\n", iam, subname);
      return 1;
   }
   if (dbg_lvl)
      printf ("(%d) %s: reading vmix_matrix vars from %s\n", iam, subname, circ_fname);
   for (A.kprime = 0; A.kprime < km; A.kprime++) {
      sprintf (varname, "vmix_matrix_%03d_CUR", A.kprime + 1);
      if (dbg_lvl)
         printf ("(%d) %s: reading %s from %s\n", iam, subname, varname, circ_fname);
      if (get_var_3d_double (circ_fname, varname, A.F))
         return 1;
      for_rows (term_vmix_matrix, &A);
   }
   free_3d_double (A.F);
   trace ("exiting", subname);
   return 0;
}

static int add_vmix (void)
{
   char *subname = "add_vmix";

   trace ("entering", subname);
   switch (vmix_opt) {
   case vmix_matrix_file:
      if (add_vmix_matrix_file ())
         return 1;
      break;
   case vmix_file:
      if (add_vmix_file ())
         return 1;
      break;
   case vmix_const:
      for_rows (term_vmix, NULL);
      break;
   case vmix_none:
      break;
   }
   if (dbg_lvl)
      printf ("(%d) vmix terms added\n\n", iam);
   trace ("exiting", subname);
   return 0;
}

/* ------------------------------------------------------------------ sinks and surface fluxes */

typedef struct { double ***F3; double **F2; double ****LEVELS; int other; } field_arg;

static void term_sink_const (const row_ctx *R, double *v, void *arg)
{
   (void) arg;
   V (SELF) += -year_cnt * per_tracer_opt[R->t].sink_rate;
}

static void term_sink_const_shallow (const row_ctx *R, double *v, void *arg)
{
   (void) arg;
   if (z_t[R->k] < per_tracer_opt[R->t].sink_depth)
      V (SELF) += -year_cnt * per_tracer_opt[R->t].sink_rate;
}

static void term_sink_file (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   V (SELF) += -year_cnt * A->F3[R->k][R->j][R->i];
}

/* decay-type sinks: rates are per year, applied over year_cnt years (reference :3059-3131) */
static int add_sink_pure_diag (void)
{
   char *subname = "add_sink_pure_diag";
   field_arg A;

   trace ("entering", subname);
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      switch (per_tracer_opt[t].sink_opt) {
      case sink_none:
      case sink_generic_tracer:
         break;
      case sink_const:
         for_rows_of_tracer (t, term_sink_const, NULL);
         if (dbg_lvl)
            printf ("(%d) sink const (%e) added for tracer %d\n\n", iam, per_tracer_opt[t].sink_rate, t);
         break;
      case sink_const_shallow:
         for_rows_of_tracer (t, term_sink_const_shallow, NULL);
         if (dbg_lvl)
            printf ("(%d) sink const shallow (%e,%e) added for tracer %d\n\n", iam, per_tracer_opt[t].sink_depth,
                    per_tracer_opt[t].sink_rate, t);
         break;
      case sink_file:
         if (dbg_lvl)
            printf ("(%d) %s: reading %s from %s\n", iam, subname, per_tracer_opt[t].sink_field_name, tracer_fname);
         if ((A.F3 = read_3d (tracer_fname, per_tracer_opt[t].sink_field_name, 0)) == NULL)
            return 1;
         for_rows_of_tracer (t, term_sink_file, &A);
         free_3d_double (A.F3);
         if (dbg_lvl)
            printf ("(%d) file sink (%s,%s) added for tracer %d\n\n", iam, tracer_fname, per_tracer_opt[t].sink_field_name, t);
         break;
      }
   }
   trace ("exiting", subname);
   return 0;
}

static void term_generic_same_level (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   V (SELF) += delta_t * A->F3[R->k][R->j][R->i];
}

static void term_generic_shallower (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   int n = R->sink0;

   for (int k2 = R->sink_cnt - 1; k2 >= 0; k2--, n++)
      if (A->LEVELS[k2] != NULL)
         v[n] += delta_t * A->LEVELS[k2][R->k][R->j][R->i];
}

/* linearised source/sink of a generic tracer: d_J_X_d_X couples a cell to itself,
 * d_J_X_d_X_k_NN couples every cell at or below level NN to level NN of its own column
 * (light / export dependencies; reference src/matrix.c:3135-3270) */
static int add_sink_generic_tracer (void)
{
   char *subname = "add_sink_generic_tracer";
   field_arg A;
   int var_exists;

   trace ("entering", subname);
   if ((A.LEVELS = (double ****) calloc ((size_t) km, sizeof (double ***))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for SINK_RATE_FIELDS_SHALLOWER\n", iam, subname);
      return 1;
   }
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      char *name;
      char *field_name;
      int kmax;

      if (per_tracer_opt[t].sink_opt != sink_generic_tracer)
         continue;
      name = per_tracer_opt[t].sink_generic_tracer_name;
      kmax = generic_sink_kmax (t);
      if ((field_name = (char *) malloc (13 + 2 * strlen (name))) == NULL) {
         fprintf (stderr, "(%d) malloc failed in %s for field_name for tracer %s\n", iam, subname, name);
         return 1;
      }
      sprintf (field_name, "d_J_%s_d_%s", name, name);
      if (var_exists_in_file (tracer_fname, field_name, &var_exists)) {
         fprintf (stderr, "(%d) var_exists_in_file failed in %s for field_name %s for tracer %s\n", iam, subname, field_name, name);
         return 1;
      }
      if (var_exists) {
         if (dbg_lvl)
            printf ("(%d) %s: reading %s from %s\n", iam, subname, field_name, tracer_fname);
         if ((A.F3 = read_3d (tracer_fname, field_name, 0)) == NULL)
            return 1;
         for_rows_of_tracer (t, term_generic_same_level, &A);
         free_3d_double (A.F3);
      } else if (dbg_lvl)
         printf ("(%d) %s: %s does not exist in %s\n", iam, subname, field_name, tracer_fname);

      for (int k2 = 0; k2 <= kmax && k2 < km; k2++) {
         sprintf (field_name, "d_J_%s_d_%s_k_%02d", name, name, k2 + 1);
         if (var_exists_in_file (tracer_fname, field_name, &var_exists)) {
            fprintf (stderr, "(%d) var_exists_in_file failed in %s for field_name %s for tracer %s\n", iam, subname, field_name, name);
            return 1;
         }
         if (var_exists) {
            if (dbg_lvl)
               printf ("(%d) %s: reading %s from %s\n", iam, subname, field_name, tracer_fname);
            if ((A.LEVELS[k2] = read_3d (tracer_fname, field_name, 0)) == NULL)
               return 1;
         } else {
            if (dbg_lvl)
               printf ("(%d) %s: %s does not exist in %s\n", iam, subname, field_name, tracer_fname);
            A.LEVELS[k2] = NULL;
         }
      }
      for_rows_of_tracer (t, term_generic_shallower, &A);
      for (int k2 = 0; k2 < km; k2++)
         if (A.LEVELS[k2] != NULL) {
            free_3d_double (A.LEVELS[k2]);
            A.LEVELS[k2] = NULL;
         }
      free (field_name);
      if (dbg_lvl)
         printf ("(%d) generic tracer sink added for tracer %d, %s\n\n", iam, t, name);
   }
   free (A.LEVELS);
   trace ("exiting", subname);
   return 0;
}

/* slot of tracer t2's copy of the cell inside row R (t2 != R->t) */
static int other_tracer_slot (const row_ctx *R, int t2)
{
   return R->other0 + ((t2 < R->t) ? t2 : t2 - 1);
}

static void term_coupled_sink (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   v[other_tracer_slot (R, A->other)] += delta_t * A->F3[R->k][R->j][R->i];
}

static void term_coupled_sf (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   if (R->k == 0)
      v[other_tracer_slot (R, A->other)] += delta_t * A->F2[R->j][R->i] / dz[0];
}

/* same-cell coupling between the tracers of a coupled pair: interior sources d_J_A_d_B
 * (both pairs) and surface fluxes d_SF_A_d_B (DIC/ALK only, as in the reference, whose
 * switch has no PO4/DOP case; reference src/matrix.c:3274-3384, 3508-3617) */
static int add_coupled_tracers (int surface_flux)
{
   char *subname = surface_flux ? "add_sf_coupled_tracers" : "add_sink_coupled_tracers";
   char **tracer_names = NULL;
   field_arg A;
   int var_exists;

   trace ("entering", subname);
   switch (coupled_tracer_opt) {
   case coupled_tracer_none:
      break;
   case coupled_tracer_OCMIP_BGC_PO4_DOP:
      if (!surface_flux)
         tracer_names = OCMIP_BGC_PO4_DOP_names;
      break;
   case coupled_tracer_DIC_SHADOW_ALK_SHADOW:
      tracer_names = DIC_SHADOW_ALK_SHADOW_names;
      break;
   }
   if (tracer_names != NULL)
      for (int t = 0; t < coupled_tracer_cnt; t++)
         for (int t2 = 0; t2 < coupled_tracer_cnt; t2++) {
            char *field_name;

            if (t2 == t)
               continue;
            if ((field_name = (char *) malloc (9 + strlen (tracer_names[t]) + strlen (tracer_names[t2]))) == NULL) {
               fprintf (stderr, "(%d) malloc failed in %s for field_name\n", iam, subname);
               return 1;
            }
            sprintf (field_name, surface_flux ? "d_SF_%s_d_%s" : "d_J_%s_d_%s", tracer_names[t], tracer_names[t2]);
            if (var_exists_in_file (tracer_fname, field_name, &var_exists)) {
               fprintf (stderr, "(%d) var_exists_in_file failed in %s for field_name %s\n", iam, subname, field_name);
               return 1;
            }
            if (var_exists) {
               if (dbg_lvl)
                  printf ("(%d) %s: reading %s from %s\n", iam, subname, field_name, tracer_fname);
               A.other = t2;
               if (surface_flux) {
                  if ((A.F2 = read_2d (tracer_fname, field_name, 0)) == NULL)
                     return 1;
                  for_rows_of_tracer (t, term_coupled_sf, &A);
                  free_2d_double (A.F2);
               } else {
                  if ((A.F3 = read_3d (tracer_fname, field_name, 0)) == NULL)
                     return 1;
                  for_rows_of_tracer (t, term_coupled_sink, &A);
                  free_3d_double (A.F3);
               }
            } else if (dbg_lvl)
               printf ("(%d) %s: %s does not exist in %s\n", iam, subname, field_name, tracer_fname);
            free (field_name);
         }
   trace ("exiting", subname);
   return 0;
}

static void term_pv (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   if (R->k == 0)
      V (SELF) -= A->F2[R->j][R->i] / dz[0] * delta_t;
}

static void term_d_SF (const row_ctx *R, double *v, void *arg)
{
   field_arg *A = (field_arg *) arg;
   if (R->k == 0)
      V (SELF) += A->F2[R->j][R->i] / dz[0] * delta_t;
}

/* surface exchange: piston velocity (a loss) and d(surface flux)/d(tracer), both spread
 * over the top layer (reference src/matrix.c:3388-3504) */
static int add_surface_2d (int is_pv)
{
   char *subname = is_pv ? "add_pv" : "add_d_SF_d_TRACER";
   field_arg A;

   trace ("entering", subname);
   for (int t = 0; t < coupled_tracer_cnt; t++) {
      char *field = is_pv ? per_tracer_opt[t].pv_field_name : per_tracer_opt[t].d_SF_d_TRACER_field_name;

      if (field == NULL)
         continue;
      if (tracer_fname == NULL) {
         fprintf (stderr, "(%d) %s:tracer_fname not specified for tracer %s %s\n", iam, subname, is_pv ? "pv" : "d_SF_d_TRACER", field);
         return 1;
      }
      if (dbg_lvl)
         printf (is_pv ? "(%d) %s: reading %s for piston velocity from %s\n" : "(%d) %s: reading %s from %s\n", iam, subname, field, tracer_fname);
      if ((A.F2 = read_2d (tracer_fname, field, 0)) == NULL)
         return 1;
      for_rows_of_tracer (t, is_pv ? term_pv : term_d_SF, &A);
      free_2d_double (A.F2);
   }
   if (dbg_lvl)
      printf (is_pv ? "(%d) pv terms added\n\n" : "(%d) d_SF_d_TRACER terms added\n\n", iam);
   trace ("exiting", subname);
   return 0;
}

/* ------------------------------------------------------------------ clean-up passes */

/* a column may appear twice in a row's pattern (east == west on a 2-wide periodic grid, the
 * vmix-matrix / generic-sink runs repeat the 7-point columns): fold later copies into the
 * first and zero them (reference src/matrix.c:3621-3650) */
static void sum_dup_vals (void)
{
   char *subname = "sum_dup_vals";
   long dup_cnt = 0;

   trace ("entering", subname);
#pragma omp parallel for schedule(static) reduction(+:dup_cnt)
   for (int row = 0; row < flat_len; row++)
      for (int_t c = rowptr[row]; c < rowptr[row + 1]; c++)
         for (int_t d = c + 1; d < rowptr[row + 1]; d++)
            if (colind[d] == colind[c]) {
               nzval_row_wise[c] += nzval_row_wise[d];
               nzval_row_wise[d] = 0.0;
               dup_cnt++;
            }
   if (dbg_lvl)
      printf ("(%d) subname = %s, dup_cnt = %ld\n", iam, subname, dup_cnt);
   trace ("exiting", subname);
}

static void strip_matrix_zeros (void)
{
   char *subname = "strip_matrix_zeros";
   int_t all = 0, kept = 0;

   trace ("entering", subname);
   for (int row = 0; row < flat_len; row++) {
      for (; all < rowptr[row + 1]; all++)
         if (nzval_row_wise[all] != 0.0) {
            nzval_row_wise[kept] = nzval_row_wise[all];
            colind[kept] = colind[all];
            kept++;
         }
      rowptr[row + 1] = kept;
   }
   if (dbg_lvl)
      printf ("(%d) subname = %s, nnz_pre = %d, nnz_new = %d\n", iam, subname, nnz, (int) kept);
   nnz = (int) kept;
   trace ("exiting", subname);
}

/* report only, like the reference (src/matrix.c:3693-3728): the solver refuses such a file */
static void check_matrix_diag (void)
{
   char *subname = "check_matrix_diag";

   trace ("entering", subname);
   for (int row = 0; row < flat_len; row++) {
      int diag_found = 0;

      for (int_t c = rowptr[row]; c < rowptr[row + 1]; c++)
         if (colind[c] == row) {
            diag_found = 1;
            if (nzval_row_wise[c] == 0.0)
               printf ("(%d) subname = %s, zero on diagonal, flat_ind = %d, flat_ind = %d, colind = %lld\n", iam, subname, row, row, (long long) colind[c]);
         }
      if (!diag_found) {
         printf ("(%d) subname = %s, no diagonal found, flat_ind = %d, flat_ind = %d, colind = ", iam, subname, row, row);
         for (int_t c = rowptr[row]; c < rowptr[row + 1]; c++)
            printf ("(%d)  %lld", iam, (long long) colind[c]);
         putchar ('\n');
      }
   }
   trace ("exiting", subname);
}

static void sort_cols_all_rows (void)
{
   char *subname = "sort_cols_all_rows";

   trace ("entering", subname);
#pragma omp parallel for schedule(static)
   for (int row = 0; row < flat_len; row++) {
      int_t *c = colind + rowptr[row];
      double *x = nzval_row_wise + rowptr[row];
      int len = (int) (rowptr[row + 1] - rowptr[row]);

      for (int a = 1; a < len; a++) {          /* rows are short: insertion sort */
         int_t key = c[a];
         double val = x[a];
         int b = a - 1;
         for (; b >= 0 && c[b] > key; b--) {
            c[b + 1] = c[b];
            x[b + 1] = x[b];
         }
         c[b + 1] = key;
         x[b + 1] = val;
      }
   }
   trace ("exiting", subname);
}

/* ------------------------------------------------------------------ driver and writer */

int gen_sparse_matrix (double day_cnt)
{
   char *subname = "gen_sparse_matrix";

   delta_t = 60.0 * 60.0 * 24.0 * day_cnt;
   year_cnt = day_cnt / 365.0;

   trace ("entering", subname);
   if (init_matrix ())
      return 1;
   /* advection first: adv_enforce_divfree rewrites the diagonal from what is there */
   if (add_adv ())
      return 1;
   if (l_adv_enforce_divfree)
      for_rows (term_divfree, NULL);
   if (add_hmix ())
      return 1;
   if (add_vmix ())
      return 1;
   if (add_sink_pure_diag ())
      return 1;
   if (add_sink_generic_tracer ())
      return 1;
   if (add_coupled_tracers (0))
      return 1;
   if (add_surface_2d (1))
      return 1;
   if (add_surface_2d (0))
      return 1;
   if (add_coupled_tracers (1))
      return 1;
   {
      double t0 = now_s ();
      sum_dup_vals ();
      strip_matrix_zeros ();
      check_matrix_diag ();
      sort_cols_all_rows ();
      if (dbg_lvl)
         printf ("(%d) %s: %.2f s reading fields, %.2f s in row passes, %.2f s folding / stripping / sorting\n", iam, subname,
                 t_read, t_rows, now_s () - t0);
   }
   trace ("exiting", subname);
   return 0;
}

int put_sparse_matrix (char *fname)
{
   char *subname = "put_sparse_matrix";
   nc3_file *f;
   int status;
   int dimids[1] = { 0 };
   int nnz_dimid, flat_len_p1_dimid;

   trace ("entering", subname);
   if ((status = nc3_open (fname, 1, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   if ((status = nc3_redef (f)))
      return handle_nc_error (subname, "nc_redef", fname, status);
   nc3_set_fill (f, 0);      /* every new variable is written in full below: skip the GB-sized pre-fill */
   if ((status = nc3_def_dim (f, "nnz", (size_t) nnz, &nnz_dimid)))
      return handle_nc_error (subname, "nc_def_dim", "nnz", status);
   if ((status = nc3_def_dim (f, "flat_len_p1", (size_t) flat_len + 1, &flat_len_p1_dimid)))
      return handle_nc_error (subname, "nc_def_dim", "flat_len_p1", status);

   if ((status = nc3_def_var (f, "coupled_tracer_cnt", NC3_INT, 0, dimids, NULL)))
      return handle_nc_error (subname, "nc_def_var", "coupled_tracer_cnt", status);
   dimids[0] = nnz_dimid;
   if ((status = nc3_def_var (f, "nzval_row_wise", NC3_DOUBLE, 1, dimids, NULL)))
      return handle_nc_error (subname, "nc_def_var", "nzval_row_wise", status);
   if ((status = nc3_def_var (f, "colind", NC3_INT, 1, dimids, NULL)))
      return handle_nc_error (subname, "nc_def_var", "colind", status);
   dimids[0] = flat_len_p1_dimid;
   if ((status = nc3_def_var (f, "rowptr", NC3_INT, 1, dimids, NULL)))
      return handle_nc_error (subname, "nc_def_var", "rowptr", status);
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);

   if (put_var_1d_int (fname, "coupled_tracer_cnt", &coupled_tracer_cnt))
      return 1;
   if (put_var_1d_double (fname, "nzval_row_wise", nzval_row_wise))
      return 1;
   if (put_var_1d_int (fname, "colind", colind))          /* int_t is int here: no staging copy */
      return 1;
   if (put_var_1d_int (fname, "rowptr", rowptr))
      return 1;
   trace ("exiting", subname);
   return 0;
}
