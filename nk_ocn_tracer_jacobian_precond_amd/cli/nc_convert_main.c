/* nc_convert -- NetCDF <--> flat binary converters for the matrix file and for tracer fields.
 *
 * The reference's TODO (TODO:4-9) asks for "matrix nc <--> binary" and "vector nc <--> binary" conversion tools so that
 * tool chains that speak flat binary can feed gen_A / solve_AB*.  Host only (libnkp_host); NetCDF classic files are read by
 * the build's own codec, netCDF-4 files through libhdf5 when one can be loaded.
 *
 *   nc_convert matrix2bin matrix.nc matrix.bin     everything the solvers read of a matrix file
 *   nc_convert bin2matrix matrix.bin matrix.nc     ... back into a CDF-2 file solve_ABglobal / solve_ABdist accept
 *   nc_convert var2bin file.nc VAR out.bin         a numeric variable, whole, as native-endian float64 in storage order
 *   nc_convert bin2var in.bin file.nc VAR          overwrites VAR in place (element count must agree)
 *
 * Binary matrix layout (native endian): int32 header[10] = { 'NKPM', 1, coupled_tracer_cnt, flat_len, nnz, imt, jmt, km,
 * tracer_state_len, 0 }, int32 rowptr[flat_len + 1], int32 colind[nnz], float64 nzval_row_wise[nnz],
 * int32 tracer_state_ind_to_i / _j / _k [tracer_state_len] each.  Indices are 0-based like the file's.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../host/nc3_codec.h"
#include "../host/nkp_host.h"

int dbg_lvl = 0, iam = 0;

#define NKPM_MAGIC 0x4D504B4E        /* "NKPM" */

static int fail (const char *what, const char *path)
{
   fprintf (stderr, "(%d) nc_convert: %s %s\n", iam, what, path);
   return EXIT_FAILURE;
}

static int matrix2bin (char *nc, char *bin)
{
   if (get_sparse_matrix (nc) || get_ind_maps (nc)) return EXIT_FAILURE;
   FILE *f = fopen (bin, "wb");
   if (!f) return fail ("cannot create", bin);
   const int32_t hdr[10] = { NKPM_MAGIC, 1, coupled_tracer_cnt, flat_len, nnz, imt, jmt, km, tracer_state_len, 0 };
   int ok = fwrite (hdr, sizeof hdr, 1, f) == 1 && fwrite (rowptr, sizeof (int32_t), (size_t) flat_len + 1, f) == (size_t) flat_len + 1 &&
            fwrite (colind, sizeof (int32_t), (size_t) nnz, f) == (size_t) nnz && fwrite (nzval_row_wise, sizeof (double), (size_t) nnz, f) == (size_t) nnz;
   for (int c = 0; c < 3 && ok; c++)
      for (int s = 0; s < tracer_state_len && ok; s++) {
         const int32_t v = c == 0 ? tracer_state_ind_to_int3[s].i : c == 1 ? tracer_state_ind_to_int3[s].j : tracer_state_ind_to_int3[s].k;
         ok = fwrite (&v, sizeof v, 1, f) == 1;
      }
   if (fclose (f) || !ok) return fail ("short write to", bin);
   free_sparse_matrix ();
   free_ind_maps ();
   return EXIT_SUCCESS;
}

static int bin2matrix (char *bin, char *nc)
{
   FILE *f = fopen (bin, "rb");
   if (!f) return fail ("cannot open", bin);
   int32_t hdr[10];
   if (fread (hdr, sizeof hdr, 1, f) != 1 || hdr[0] != NKPM_MAGIC || hdr[1] != 1) { fclose (f); return fail ("not an nc_convert matrix file:", bin); }
   coupled_tracer_cnt = hdr[2]; flat_len = hdr[3]; nnz = hdr[4]; imt = hdr[5]; jmt = hdr[6]; km = hdr[7]; tracer_state_len = hdr[8];
   if (coupled_tracer_cnt < 1 || flat_len < 0 || nnz < 0 || imt < 1 || jmt < 1 || km < 1 || tracer_state_len < 0 ||
       (long long) coupled_tracer_cnt * tracer_state_len != flat_len) { fclose (f); return fail ("inconsistent header in", bin); }
   rowptr = (int_t *) malloc (((size_t) flat_len + 1) * sizeof (int_t));
   colind = (int_t *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (int_t));
   nzval_row_wise = (double *) malloc ((size_t) (nnz ? nnz : 1) * sizeof (double));
   tracer_state_ind_to_int3 = (int3 *) malloc ((size_t) (tracer_state_len ? tracer_state_len : 1) * sizeof (int3));
   int32_t *col = (int32_t *) malloc ((size_t) (tracer_state_len ? tracer_state_len : 1) * sizeof (int32_t));
   int3_to_tracer_state_ind = malloc_3d_int (km, jmt, imt);
   if (!rowptr || !colind || !nzval_row_wise || !tracer_state_ind_to_int3 || !col || !int3_to_tracer_state_ind) { fclose (f); return fail ("out of memory reading", bin); }
   int ok = fread (rowptr, sizeof (int32_t), (size_t) flat_len + 1, f) == (size_t) flat_len + 1 && fread (colind, sizeof (int32_t), (size_t) nnz, f) == (size_t) nnz &&
            fread (nzval_row_wise, sizeof (double), (size_t) nnz, f) == (size_t) nnz;
   for (int c = 0; c < 3 && ok; c++) {
      ok = fread (col, sizeof (int32_t), (size_t) tracer_state_len, f) == (size_t) tracer_state_len;
      for (int s = 0; s < tracer_state_len && ok; s++) {
         if (c == 0) tracer_state_ind_to_int3[s].i = col[s];
         else if (c == 1) tracer_state_ind_to_int3[s].j = col[s];
         else tracer_state_ind_to_int3[s].k = col[s];
      }
   }
   fclose (f);
   free (col);
   if (!ok) return fail ("short read from", bin);
   for (int k = 0; k < km; k++)
      for (int j = 0; j < jmt; j++)
         for (int i = 0; i < imt; i++) int3_to_tracer_state_ind[k][j][i] = -1;
   for (int s = 0; s < tracer_state_len; s++) {
      const int3 p = tracer_state_ind_to_int3[s];
      if (p.i < 0 || p.i >= imt || p.j < 0 || p.j >= jmt || p.k < 0 || p.k >= km) return fail ("index map out of the grid in", bin);
      int3_to_tracer_state_ind[p.k][p.j][p.i] = s;
   }
   /* the grid dimensions the readers look up (src/grid.c:50-69), then the reference's own two writers */
   nc3_file *g;
   int status, dimid;
   if ((status = nc3_create (nc, 2, &g))) return handle_nc_error ("bin2matrix", "nc_create", nc, status);
   if ((status = nc3_def_dim (g, "nlon", (size_t) imt, &dimid)) || (status = nc3_def_dim (g, "nlat", (size_t) jmt, &dimid)) ||
       (status = nc3_def_dim (g, "z_t", (size_t) km, &dimid)))
      return handle_nc_error ("bin2matrix", "nc_def_dim", nc, status);
   if ((status = nc3_close (g))) return handle_nc_error ("bin2matrix", "nc_close", nc, status);
   if (put_ind_maps (nc) || put_sparse_matrix (nc)) return EXIT_FAILURE;
   return EXIT_SUCCESS;
}

static int var2bin (char *nc, char *var, char *bin)
{
   size_t n = 0;
   if (nkp_var_nelems (nc, var, &n)) return EXIT_FAILURE;
   double *buf = (double *) malloc ((n ? n : 1) * sizeof (double));
   if (!buf) return fail ("out of memory for", var);
   if (get_var_1d_double (nc, var, buf)) return EXIT_FAILURE;          /* whole variable, any numeric type, as float64 */
   FILE *f = fopen (bin, "wb");
   if (!f) return fail ("cannot create", bin);
   const int ok = fwrite (buf, sizeof (double), n, f) == n;
   if (fclose (f) || !ok) return fail ("short write to", bin);
   free (buf);
   return EXIT_SUCCESS;
}

static int bin2var (char *bin, char *nc, char *var)
{
   size_t n = 0;
   if (nkp_var_nelems (nc, var, &n)) return EXIT_FAILURE;
   double *buf = (double *) malloc ((n ? n : 1) * sizeof (double));
   if (!buf) return fail ("out of memory for", var);
   FILE *f = fopen (bin, "rb");
   if (!f) return fail ("cannot open", bin);
   const size_t got = fread (buf, sizeof (double), n, f);
   const int extra = fgetc (f) != EOF;
   fclose (f);
   if (got != n || extra) {
      fprintf (stderr, "(%d) nc_convert: %s holds %s float64 values than %s has elements (%zu)\n", iam, bin, extra ? "more" : "fewer", var, n);
      return EXIT_FAILURE;
   }
   if (put_var_1d_double (nc, var, buf)) return EXIT_FAILURE;
   free (buf);
   return EXIT_SUCCESS;
}

int main (int argc, char **argv)
{
   if (argc == 4 && !strcmp (argv[1], "matrix2bin")) return matrix2bin (argv[2], argv[3]);
   if (argc == 4 && !strcmp (argv[1], "bin2matrix")) return bin2matrix (argv[2], argv[3]);
   if (argc == 5 && !strcmp (argv[1], "var2bin")) return var2bin (argv[2], argv[3], argv[4]);
   if (argc == 5 && !strcmp (argv[1], "bin2var")) return bin2var (argv[2], argv[3], argv[4]);
   fprintf (stderr, "usage: nc_convert matrix2bin matrix.nc matrix.bin | bin2matrix matrix.bin matrix.nc | var2bin file.nc VAR out.bin | bin2var in.bin file.nc VAR\n");
   return EXIT_FAILURE;
}
