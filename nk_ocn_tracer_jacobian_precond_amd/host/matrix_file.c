/* Readers for the matrix file gen_A writes: grid dimensions, index maps and the CSR arrays.
 * Same names, globals and failure behaviour as the reference (src/grid.c:33-86,
 * src/matrix.c:373-464, 3943-4070); schema in SURVEY.md section 3.3.  Differences, all
 * deliberate: no SuperLU int_t temp copies (the file type IS int32), none of the six
 * coef_ind_* work arrays the reference allocates but never uses here (matrix.c:714-738),
 * and the arrays are validated (monotone rowptr, in-range sorted columns) before use
 * because they are about to be handed to GPU kernels that trust them. */
#include <stdio.h>
#include <stdlib.h>

#include "nc3_codec.h"
#include "nkp_host.h"

int imt, jmt, km;

int tracer_state_len;
int ***int3_to_tracer_state_ind = NULL;
int3 *tracer_state_ind_to_int3 = NULL;

int coupled_tracer_cnt;
int flat_len;
int nnz;
double *nzval_row_wise = NULL;
int_t *colind = NULL;
int_t *rowptr = NULL;

static void trace (const char *what, const char *subname)
{
   if (dbg_lvl > 1) {
      printf ("(%d) %s %s\n", iam, what, subname);
      fflush (stdout);
   }
}

/* read several dimension lengths with one open/close; names[i] -> lens[i] */
static int read_dimlens (char *subname, char *fname, int n, char **names, size_t *lens)
{
   nc3_file *f;
   int status;

   if ((status = nc3_open (fname, 0, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   for (int d = 0; d < n; d++)
      if ((status = nc3_inq_dimlen (f, names[d], &lens[d]))) {
         nc3_close (f);
         return handle_nc_error (subname, "nc_inq_dimid", names[d], status);
      }
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);
   return 0;
}

int get_grid_dims (char *fname)
{
   char *subname = "get_grid_dims";
   char *names[3] = { "nlon", "nlat", "z_t" };
   size_t lens[3];

   trace ("entering", subname);
   if (read_dimlens (subname, fname, 3, names, lens))
      return 1;
   imt = (int) lens[0];
   jmt = (int) lens[1];
   km = (int) lens[2];
   if (dbg_lvl && iam == 0) {
      printf ("(%d) imt = %d\n", iam, imt);
      printf ("(%d) jmt = %d\n", iam, jmt);
      printf ("(%d) km  = %d\n", iam, km);
   }
   trace ("exiting", subname);
   return 0;
}

int get_ind_maps (char *fname)
{
   char *subname = "get_ind_maps";
   char *names[1] = { "tracer_state_len" };
   char *ijk_vars[3] = { "tracer_state_ind_to_i", "tracer_state_ind_to_j", "tracer_state_ind_to_k" };
   size_t len;
   int *tmp;

   trace ("entering", subname);
   if (get_grid_dims (fname))
      return 1;
   if (read_dimlens (subname, fname, 1, names, &len))
      return 1;
   tracer_state_len = (int) len;
   if (dbg_lvl && iam == 0)
      printf ("(%d) %s: tracer_state_len = %d\n", iam, subname, tracer_state_len);

   if ((int3_to_tracer_state_ind = malloc_3d_int (km, jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for int3_to_tracer_state_ind\n", iam, subname);
      return 1;
   }
   if ((tracer_state_ind_to_int3 = (int3 *) malloc ((size_t) (tracer_state_len ? tracer_state_len : 1) * sizeof (int3))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for tracer_state_ind_to_int3\n", iam, subname);
      return 1;
   }
   if (get_var_3d_int (fname, "int3_to_tracer_state_ind", int3_to_tracer_state_ind))
      return 1;

   if ((tmp = (int *) malloc ((size_t) (tracer_state_len ? tracer_state_len : 1) * sizeof (int))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for tracer_state_ind_to_ijk\n", iam, subname);
      return 1;
   }
   for (int c = 0; c < 3; c++) {
      if (get_var_1d_int (fname, ijk_vars[c], tmp)) {
         free (tmp);
         return 1;
      }
      int *dst = (c == 0) ? &tracer_state_ind_to_int3[0].i : (c == 1) ? &tracer_state_ind_to_int3[0].j : &tracer_state_ind_to_int3[0].k;
      for (int s = 0; s < tracer_state_len; s++)
         dst[3 * (size_t) s] = tmp[s];      /* int3 is three packed ints */
   }
   free (tmp);

   for (int s = 0; s < tracer_state_len; s++) {
      int3 c = tracer_state_ind_to_int3[s];
      if (c.i < 0 || c.i >= imt || c.j < 0 || c.j >= jmt || c.k < 0 || c.k >= km) {
         fprintf (stderr, "(%d) %s: tracer_state_ind %d maps outside the grid (i=%d j=%d k=%d)\n", iam, subname, s, c.i, c.j, c.k);
         return 1;
      }
   }
   trace ("exiting", subname);
   return 0;
}

void free_ind_maps (void)
{
   free_3d_int (int3_to_tracer_state_ind);
   free (tracer_state_ind_to_int3);
   int3_to_tracer_state_ind = NULL;
   tracer_state_ind_to_int3 = NULL;
}

int get_sparse_matrix (char *fname)
{
   char *subname = "get_sparse_matrix";
   char *names[2] = { "nnz", "flat_len_p1" };
   size_t lens[2];

   trace ("entering", subname);
   if (read_dimlens (subname, fname, 2, names, lens))
      return 1;
   if (lens[0] > 2147483647u || lens[1] == 0 || lens[1] > 2147483647u) {
      fprintf (stderr, "(%d) %s: nnz=%zu flat_len_p1=%zu outside the int32 schema\n", iam, subname, lens[0], lens[1]);
      return 1;
   }
   nnz = (int) lens[0];
   flat_len = (int) lens[1] - 1;

   if (get_var_1d_int (fname, "coupled_tracer_cnt", &coupled_tracer_cnt))
      return 1;
   if (dbg_lvl && iam == 0) {
      printf ("(%d) %s: coupled_tracer_cnt = %d\n", iam, subname, coupled_tracer_cnt);
      printf ("(%d) %s: nnz = %d\n", iam, subname, nnz);
      printf ("(%d) %s: flat_len = %d\n", iam, subname, flat_len);
   }

   nzval_row_wise = (double *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (double));
   colind = (int_t *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (int_t));
   rowptr = (int_t *) malloc ((size_t) (flat_len + 1) * sizeof (int_t));
   if (!nzval_row_wise || !colind || !rowptr) {
      fprintf (stderr, "(%d) malloc failed in %s for the CSR arrays\n", iam, subname);
      return 1;
   }
   if (get_var_1d_double (fname, "nzval_row_wise", nzval_row_wise))
      return 1;
   if (get_var_1d_int (fname, "colind", colind))
      return 1;
   if (get_var_1d_int (fname, "rowptr", rowptr))
      return 1;

   /* structural validation: 0-based, monotone, last pointer == nnz, sorted in-range columns */
   if (rowptr[0] != 0 || rowptr[flat_len] != nnz) {
      fprintf (stderr, "(%d) %s: rowptr[0]=%d rowptr[flat_len]=%d inconsistent with nnz=%d\n", iam, subname, rowptr[0], rowptr[flat_len], nnz);
      return 1;
   }
   for (int r = 0; r < flat_len; r++) {
      if (rowptr[r + 1] < rowptr[r]) {
         fprintf (stderr, "(%d) %s: rowptr decreases at row %d\n", iam, subname, r);
         return 1;
      }
      for (int e = rowptr[r]; e < rowptr[r + 1]; e++)
         if (colind[e] < 0 || colind[e] >= flat_len || (e > rowptr[r] && colind[e] <= colind[e - 1])) {
            fprintf (stderr, "(%d) %s: row %d has an out-of-range or unsorted column index %d\n", iam, subname, r, colind[e]);
            return 1;
         }
   }
   trace ("exiting", subname);
   return 0;
}

/* dims, coupled_tracer_cnt and rowptr: what every rank of the row-distributed flavour needs of the matrix before it knows
 * its rows (reference: rank 0 reads everything and sends rowptr slices, src/solve_ABdist.c:141-175) */
int get_sparse_matrix_header (char *fname)
{
   char *subname = "get_sparse_matrix_header";
   char *names[2] = { "nnz", "flat_len_p1" };
   size_t lens[2];

   trace ("entering", subname);
   if (read_dimlens (subname, fname, 2, names, lens))
      return 1;
   if (lens[0] > 2147483647u || lens[1] == 0 || lens[1] > 2147483647u) {
      fprintf (stderr, "(%d) %s: nnz=%zu flat_len_p1=%zu outside the int32 schema\n", iam, subname, lens[0], lens[1]);
      return 1;
   }
   nnz = (int) lens[0];
   flat_len = (int) lens[1] - 1;
   if (get_var_1d_int (fname, "coupled_tracer_cnt", &coupled_tracer_cnt))
      return 1;
   rowptr = (int_t *) malloc ((size_t) (flat_len + 1) * sizeof (int_t));
   if (!rowptr) {
      fprintf (stderr, "(%d) malloc failed in %s for rowptr\n", iam, subname);
      return 1;
   }
   if (get_var_1d_int (fname, "rowptr", rowptr))
      return 1;
   if (rowptr[0] != 0 || rowptr[flat_len] != nnz) {
      fprintf (stderr, "(%d) %s: rowptr[0]=%d rowptr[flat_len]=%d inconsistent with nnz=%d\n", iam, subname, rowptr[0], rowptr[flat_len], nnz);
      return 1;
   }
   for (int r = 0; r < flat_len; r++)
      if (rowptr[r + 1] < rowptr[r]) {
         fprintf (stderr, "(%d) %s: rowptr decreases at row %d\n", iam, subname, r);
         return 1;
      }
   trace ("exiting", subname);
   return 0;
}

/* entries of rows [row0, row1) (the reference's colind / nzval slices, src/solve_ABdist.c:188-225); needs the header */
int get_sparse_matrix_rows (char *fname, int row0, int row1, int_t *colind_out, double *nzval_out)
{
   char *subname = "get_sparse_matrix_rows";
   trace ("entering", subname);
   if (!rowptr || row0 < 0 || row1 < row0 || row1 > flat_len) {
      fprintf (stderr, "(%d) %s: rows [%d, %d) outside the matrix (or get_sparse_matrix_header not called)\n", iam, subname, row0, row1);
      return 1;
   }
   const size_t e0 = (size_t) rowptr[row0], cnt = (size_t) (rowptr[row1] - rowptr[row0]);
   if (cnt) {
      if (get_vara_1d_double (fname, "nzval_row_wise", e0, cnt, nzval_out))
         return 1;
      if (get_vara_1d_int (fname, "colind", e0, cnt, colind_out))
         return 1;
   }
   for (int r = row0; r < row1; r++)
      for (int e = rowptr[r]; e < rowptr[r + 1]; e++) {
         const int_t c = colind_out[(size_t) e - e0];
         if (c < 0 || c >= flat_len || (e > rowptr[r] && c <= colind_out[(size_t) e - e0 - 1])) {
            fprintf (stderr, "(%d) %s: row %d has an out-of-range or unsorted column index %d\n", iam, subname, r, c);
            return 1;
         }
      }
   trace ("exiting", subname);
   return 0;
}

void free_sparse_matrix (void)
{
   trace ("entering", "free_sparse_matrix");
   free (rowptr);
   free (colind);
   free (nzval_row_wise);
   rowptr = colind = NULL;
   nzval_row_wise = NULL;
   trace ("exiting", "free_sparse_matrix");
}

/* ---------------------------------------------------------------- additions of this build */

int_t *nkp_column_blocks (int *nblk)
{
   int per_tracer = 0;
   for (int s = 0; s < tracer_state_len; s++)
      per_tracer += (tracer_state_ind_to_int3[s].k == 0);
   int cnt = coupled_tracer_cnt > 0 ? coupled_tracer_cnt : 1;
   int_t *start = (int_t *) malloc (((size_t) per_tracer * (size_t) cnt + 1) * sizeof (int_t));
   if (!start) return NULL;
   int b = 0;
   for (int t = 0; t < cnt; t++)
      for (int s = 0; s < tracer_state_len; s++)
         if (tracer_state_ind_to_int3[s].k == 0)
            start[b++] = t * tracer_state_len + s;
   start[b] = cnt * tracer_state_len;
   *nblk = b;
   return start;
}

void nkp_flatten_tracer (int tracer_ind, double ***field_3d, double *B)
{
   double *dst = B + (size_t) tracer_ind * (size_t) tracer_state_len;
   for (int s = 0; s < tracer_state_len; s++) {
      int3 c = tracer_state_ind_to_int3[s];
      dst[s] = field_3d[c.k][c.j][c.i];
   }
}

void nkp_unflatten_tracer (int tracer_ind, const double *B, double ***field_3d)
{
   const double *src = B + (size_t) tracer_ind * (size_t) tracer_state_len;
   for (int s = 0; s < tracer_state_len; s++) {
      int3 c = tracer_state_ind_to_int3[s];
      field_3d[c.k][c.j][c.i] = src[s];
   }
}

void nkp_rowblock_partition (int n, int nprocs, int rank, int *fst_row, int *m_loc)
{
   int base = n / nprocs;
   *fst_row = rank * base;
   *m_loc = (rank == nprocs - 1) ? n - *fst_row : base;
}

/* grid position (i, j) of every block returned by nkp_column_blocks, same order; 0 = ok */
int nkp_column_coords (int nblk, int *col_i, int *col_j)
{
   int b = 0;
   int cnt = coupled_tracer_cnt > 0 ? coupled_tracer_cnt : 1;
   for (int t = 0; t < cnt; t++)
      for (int s = 0; s < tracer_state_len; s++)
         if (tracer_state_ind_to_int3[s].k == 0) {
            if (b >= nblk) return 1;
            col_i[b] = tracer_state_ind_to_int3[s].i;
            col_j[b] = tracer_state_ind_to_int3[s].j;
            b++;
         }
   return b == nblk ? 0 : 1;
}

/* the reference's contiguous split (src/solve_ABdist.c:141-144) with every cut moved to the nearest
 * water-column boundary, so that no column straddles two ranks; blk_start has nblk + 1 entries */
void nkp_rowblock_partition_snapped (const int_t *blk_start, int nblk, int nprocs, int rank, int *fst_row, int *m_loc, int *fst_blk, int *nblk_loc)
{
   int n = blk_start[nblk];
   int cut[2];
   for (int w = 0; w < 2; w++) {
      int r = rank + w;
      if (r <= 0) { cut[w] = 0; continue; }
      if (r >= nprocs) { cut[w] = nblk; continue; }
      long target = (long) r * (n / nprocs);
      int lo = 0, hi = nblk;                          /* first block boundary >= target */
      while (lo < hi) { int mid = (lo + hi) / 2; if (blk_start[mid] < target) lo = mid + 1; else hi = mid; }
      int b = lo;
      if (b > 0 && target - blk_start[b - 1] <= blk_start[b] - target) b--;
      cut[w] = b;
   }
   if (cut[1] < cut[0]) cut[1] = cut[0];
   *fst_blk = cut[0];
   *nblk_loc = cut[1] - cut[0];
   *fst_row = blk_start[cut[0]];
   *m_loc = blk_start[cut[1]] - blk_start[cut[0]];
}
