/* Grid loader and writer of the matrix generator (reference src/grid.c:90-330): reads the
 * geometry gen_A needs from a POP history file, masks KMT, derives KMU, and writes the
 * grid part of the matrix file (dims nlon/nlat/z_t; z_t, TLONG, TLAT, masked KMT with the
 * reference's attribute texts -- dz is deliberately NOT written, as in the reference).
 * Same names, globals and failure behaviour as the reference; I/O goes through nc3_codec. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nc3_codec.h"
#include "nkp_host.h"

char *circ_fname = NULL;
char *reg_fname = NULL;

double *z_t = NULL;
double *dz = NULL;
double **TLONG = NULL;
double **TLAT = NULL;
int **KMT = NULL;
int **KMU = NULL;
double **TAREA = NULL;

static void trace (const char *what, const char *subname)
{
   if (dbg_lvl > 1) {
      printf ("(%d) %s %s\n", iam, what, subname);
      fflush (stdout);
   }
}

/* the southern- and northern-most rows must be land: every stencil reads j-1 and j+1
 * without a bounds check (reference src/grid.c:163-181, src/matrix.c:176-189) */
int nkp_check_polar_rows (const char *subname)
{
   int south_flag = 0, north_flag = 0;

   for (int i = 0; i < imt; i++) {
      if (KMT[0][i]) south_flag = 1;
      if (KMT[jmt - 1][i]) north_flag = 1;
   }
   if (south_flag)
      fprintf (stderr, "(%d) non-land found on southern-most row in %s\n", iam, subname);
   if (north_flag)
      fprintf (stderr, "(%d) non-land found on northern-most row in %s\n", iam, subname);
   return south_flag || north_flag;
}

int get_grid_info (char *circ_fname_arg, char *reg_fname_arg)
{
   char *subname = "get_grid_info";
   struct { double ***dst; char *name; } fields_2d[] = { { &TLONG, "TLONG" }, { &TLAT, "TLAT" } };

   trace ("entering", subname);
   if (get_grid_dims (circ_fname_arg))
      return 1;

   if ((z_t = (double *) malloc ((size_t) km * sizeof (double))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for z_t\n", iam, subname);
      return 1;
   }
   if (get_var_1d_double (circ_fname_arg, "z_t", z_t))
      return 1;
   if ((dz = (double *) malloc ((size_t) km * sizeof (double))) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for dz\n", iam, subname);
      return 1;
   }
   if (get_var_1d_double (circ_fname_arg, "dz", dz))
      return 1;

   for (int f = 0; f < 2; f++) {
      if ((*fields_2d[f].dst = malloc_2d_double (jmt, imt)) == NULL) {
         fprintf (stderr, "(%d) malloc failed in %s for %s\n", iam, subname, fields_2d[f].name);
         return 1;
      }
      if (get_var_2d_double (circ_fname_arg, fields_2d[f].name, *fields_2d[f].dst))
         return 1;
   }

   if ((KMT = malloc_2d_int (jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for KMT\n", iam, subname);
      return 1;
   }
   if (get_var_2d_int (circ_fname_arg, "KMT", KMT))
      return 1;
   /* negative KMT counts as land */
   for (int j = 0; j < jmt; j++)
      for (int i = 0; i < imt; i++)
         if (KMT[j][i] < 0)
            KMT[j][i] = 0;

   /* regions flagged negative in DYN_REGMASK are ignored (interior rows only) */
   if (reg_fname_arg != NULL) {
      int **DYN_REGMASK;

      if ((DYN_REGMASK = malloc_2d_int (jmt, imt)) == NULL) {
         fprintf (stderr, "(%d) malloc failed in %s for DYN_REGMASK\n", iam, subname);
         return 1;
      }
      if (get_var_2d_int (reg_fname_arg, "DYN_REGMASK", DYN_REGMASK))
         return 1;
      for (int j = 1; j < jmt - 1; j++)
         for (int i = 0; i < imt; i++)
            if (DYN_REGMASK[j][i] < 0)
               KMT[j][i] = 0;
      free_2d_int (DYN_REGMASK);
   }

   if (nkp_check_polar_rows (subname))
      return 1;

   /* KMU: depth of the velocity point at the north-east corner = min over its four T cells */
   if ((KMU = malloc_2d_int (jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for KMU\n", iam, subname);
      return 1;
   }
   for (int j = 0; j < jmt - 1; j++)
      for (int i = 0; i < imt; i++) {
         int ip1 = (i < imt - 1) ? i + 1 : 0;
         int m = KMT[j][i];
         if (KMT[j + 1][i] < m) m = KMT[j + 1][i];
         if (KMT[j][ip1] < m) m = KMT[j][ip1];
         if (KMT[j + 1][ip1] < m) m = KMT[j + 1][ip1];
         KMU[j][i] = m;
      }
   for (int i = 0; i < imt; i++)
      KMU[jmt - 1][i] = 0;

   if ((TAREA = malloc_2d_double (jmt, imt)) == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for TAREA\n", iam, subname);
      return 1;
   }
   if (get_var_2d_double (circ_fname_arg, "TAREA", TAREA))
      return 1;

   trace ("exiting", subname);
   return 0;
}

typedef struct { char *att; char *text; } text_att;

static int def_var_with_atts (char *subname, nc3_file *f, char *name, int type, int ndims, int *dimids, int natts, text_att *atts)
{
   int status, varid;

   if ((status = nc3_def_var (f, name, type, ndims, dimids, &varid)))
      return handle_nc_error (subname, "nc_def_var", name, status);
   for (int a = 0; a < natts; a++)
      if ((status = nc3_put_att_text (f, varid, atts[a].att, strlen (atts[a].text), atts[a].text)))
         return handle_nc_error (subname, "nc_put_att_text", name, status);
   return 0;
}

int put_grid_info (char *fname)
{
   char *subname = "put_grid_info";
   nc3_file *f;
   int status;
   int dimids[2];
   int nlon_dimid, nlat_dimid, z_t_dimid;
   text_att z_t_atts[] = { { "long_name", "depth from surface to midpoint of layer" }, { "units", "centimeters" }, { "positive", "down" } };
   text_att tlong_atts[] = { { "long_name", "array of t-grid longitudes" }, { "units", "degrees_east" } };
   text_att tlat_atts[] = { { "long_name", "array of t-grid latitudes" }, { "units", "degrees_north" } };
   text_att kmt_atts[] = { { "long_name", "k Index of Deepest Grid Cell on T Grid" }, { "coordinates", "TLONG TLAT" } };

   trace ("entering", subname);

   /* NC_64BIT_OFFSET in the reference = CDF-2 */
   if ((status = nc3_create (fname, 2, &f)))
      return handle_nc_error (subname, "nc_create", fname, status);

   if ((status = nc3_def_dim (f, "nlon", (size_t) imt, &nlon_dimid)))
      return handle_nc_error (subname, "nc_def_dimid", "nlon", status);
   if ((status = nc3_def_dim (f, "nlat", (size_t) jmt, &nlat_dimid)))
      return handle_nc_error (subname, "nc_def_dimid", "nlat", status);
   if ((status = nc3_def_dim (f, "z_t", (size_t) km, &z_t_dimid)))
      return handle_nc_error (subname, "nc_def_dimid", "z_t", status);

   dimids[0] = z_t_dimid;
   if (def_var_with_atts (subname, f, "z_t", NC3_DOUBLE, 1, dimids, 3, z_t_atts))
      return 1;
   dimids[0] = nlat_dimid;
   dimids[1] = nlon_dimid;
   if (def_var_with_atts (subname, f, "TLONG", NC3_DOUBLE, 2, dimids, 2, tlong_atts))
      return 1;
   if (def_var_with_atts (subname, f, "TLAT", NC3_DOUBLE, 2, dimids, 2, tlat_atts))
      return 1;
   if (def_var_with_atts (subname, f, "KMT", NC3_INT, 2, dimids, 2, kmt_atts))
      return 1;

   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);

   if (put_var_1d_double (fname, "z_t", z_t))
      return 1;
   if (put_var_2d_double (fname, "TLONG", TLONG))
      return 1;
   if (put_var_2d_double (fname, "TLAT", TLAT))
      return 1;
   if (put_var_2d_int (fname, "KMT", KMT))
      return 1;

   trace ("exiting", subname);
   return 0;
}

void free_grid_info (void)
{
   free (z_t);
   free (dz);
   free_2d_double (TLONG);
   free_2d_double (TLAT);
   free_2d_int (KMT);
   free_2d_int (KMU);
   free_2d_double (TAREA);
   z_t = dz = NULL;
   TLONG = TLAT = TAREA = NULL;
   KMT = KMU = NULL;
}
