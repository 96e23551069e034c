/* Grid part of the matrix generator: what gen_A reads from a POP history file (geometry, masked KMT, derived
 * KMU) and the grid section it writes at the head of the matrix file.  Behaviour follows the reference
 * (src/grid.c:90-330: same global names, same file schema and attribute texts, dz read but NOT written, same
 * refusal of ocean on the two polar rows); the code is organised around two tables -- the fields to load and the
 * variables to write -- and I/O goes through the file_io layer on top of nc3_codec. */
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

/* ---------------------------------------------------------------- field table */
enum field_shape { LEVELS_F64, PLANE_F64, PLANE_I32 };

typedef struct {
   const char *name;
   enum field_shape shape;
   void *slot;                 /* address of the global that receives the array */
   int from_file;              /* 1: read from the history file; 0: allocated only (derived here) */
} grid_field;

static grid_field grid_fields[] = {
   { "z_t", LEVELS_F64, &z_t, 1 },   { "dz", LEVELS_F64, &dz, 1 },
   { "TLONG", PLANE_F64, &TLONG, 1 }, { "TLAT", PLANE_F64, &TLAT, 1 },
   { "KMT", PLANE_I32, &KMT, 1 },     { "KMU", PLANE_I32, &KMU, 0 },
   { "TAREA", PLANE_F64, &TAREA, 1 },
};
#define N_GRID_FIELDS ((int) (sizeof grid_fields / sizeof grid_fields[0]))

static int load_field (const grid_field *g, char *fname, const char *who)
{
   int failed = 0;

   switch (g->shape) {
   case LEVELS_F64: {
      double *a = (double *) malloc ((size_t) km * sizeof (double));
      *(double **) g->slot = a;
      failed = (a == NULL);
      if (!failed && g->from_file && get_var_1d_double (fname, (char *) g->name, a)) return 1;
      break;
   }
   case PLANE_F64: {
      double **a = malloc_2d_double (jmt, imt);
      *(double ***) g->slot = a;
      failed = (a == NULL);
      if (!failed && g->from_file && get_var_2d_double (fname, (char *) g->name, a)) return 1;
      break;
   }
   case PLANE_I32: {
      int **a = malloc_2d_int (jmt, imt);
      *(int ***) g->slot = a;
      failed = (a == NULL);
      if (!failed && g->from_file && get_var_2d_int (fname, (char *) g->name, a)) return 1;
      break;
   }
   }
   if (failed) {
      fprintf (stderr, "(%d) malloc failed in %s for %s\n", iam, who, g->name);
      return 1;
   }
   return 0;
}

static void say (const char *what, const char *who)
{
   if (dbg_lvl > 1) {
      printf ("(%d) %s %s\n", iam, what, who);
      fflush (stdout);
   }
}

/* Every stencil reads j - 1 and j + 1 unguarded (src/matrix.c:176-189), so rows 0 and jmt - 1 must be land. */
int nkp_check_polar_rows (const char *subname)
{
   const int rows[2] = { 0, jmt - 1 };
   const char *label[2] = { "southern-most", "northern-most" };
   int bad = 0;

   for (int e = 0; e < 2; e++) {
      int wet = 0;
      for (int i = 0; i < imt && !wet; i++) wet = KMT[rows[e]][i] != 0;
      if (wet) {
         fprintf (stderr, "(%d) non-land found on %s row in %s\n", iam, label[e], subname);
         bad = 1;
      }
   }
   return bad;
}

/* land = KMT <= 0; cells of regions the optional region file flags negative are land too (interior rows) */
static int mask_depths (char *reg_file, const char *who)
{
   for (int j = 0; j < jmt; j++)
      for (int i = 0; i < imt; i++)
         if (KMT[j][i] < 0) KMT[j][i] = 0;
   if (reg_file == NULL) return 0;

   int **mask = malloc_2d_int (jmt, imt);
   if (mask == NULL) {
      fprintf (stderr, "(%d) malloc failed in %s for DYN_REGMASK\n", iam, who);
      return 1;
   }
   if (get_var_2d_int (reg_file, "DYN_REGMASK", mask)) return 1;
   for (int j = 1; j < jmt - 1; j++)
      for (int i = 0; i < imt; i++)
         if (mask[j][i] < 0) KMT[j][i] = 0;
   free_2d_int (mask);
   return 0;
}

/* a velocity point (north-east corner of T cell i, j) is as deep as the shallowest of its four T cells;
 * the northern-most row has none, the grid is periodic in i */
static void corner_depths (void)
{
   for (int j = 0; j < jmt; j++)
      for (int i = 0; i < imt; i++) {
         int depth = 0;
         if (j < jmt - 1) {
            const int e = (i + 1) % imt;
            const int four[4] = { KMT[j][i], KMT[j + 1][i], KMT[j][e], KMT[j + 1][e] };
            depth = four[0];
            for (int q = 1; q < 4; q++)
               if (four[q] < depth) depth = four[q];
         }
         KMU[j][i] = depth;
      }
}

int get_grid_info (char *circ_file, char *reg_file)
{
   const char *who = "get_grid_info";

   say ("entering", who);
   if (get_grid_dims (circ_file)) return 1;
   for (int f = 0; f < N_GRID_FIELDS; f++) {
      if (load_field (&grid_fields[f], circ_file, who)) return 1;
      /* the depth mask is final before anything derived from it is touched */
      if (grid_fields[f].slot == (void *) &KMT && (mask_depths (reg_file, who) || nkp_check_polar_rows (who))) return 1;
   }
   corner_depths ();
   say ("exiting", who);
   return 0;
}

/* ---------------------------------------------------------------- grid section of the matrix file */
typedef struct { const char *name; const char *text; } text_att;
typedef struct {
   const char *name;
   int type, ndims;
   int dims[2];                /* indices into the dimension list below */
   text_att atts[3];
} out_var;

static const char *out_dims[3] = { "nlon", "nlat", "z_t" };
static const out_var out_vars[] = {
   { "z_t", NC3_DOUBLE, 1, { 2, 0 }, { { "long_name", "depth from surface to midpoint of layer" }, { "units", "centimeters" }, { "positive", "down" } } },
   { "TLONG", NC3_DOUBLE, 2, { 1, 0 }, { { "long_name", "array of t-grid longitudes" }, { "units", "degrees_east" }, { NULL, NULL } } },
   { "TLAT", NC3_DOUBLE, 2, { 1, 0 }, { { "long_name", "array of t-grid latitudes" }, { "units", "degrees_north" }, { NULL, NULL } } },
   { "KMT", NC3_INT, 2, { 1, 0 }, { { "long_name", "k Index of Deepest Grid Cell on T Grid" }, { "coordinates", "TLONG TLAT" }, { NULL, NULL } } },
};
#define N_OUT_VARS ((int) (sizeof out_vars / sizeof out_vars[0]))

int put_grid_info (char *fname)
{
   char *who = "put_grid_info";
   const size_t dim_len[3] = { (size_t) imt, (size_t) jmt, (size_t) km };
   int dimid[3], status;
   nc3_file *f;

   say ("entering", who);
   if ((status = nc3_create (fname, 2, &f)))              /* CDF-2, the reference's NC_64BIT_OFFSET */
      return handle_nc_error (who, "nc_create", fname, status);
   for (int d = 0; d < 3; d++)
      if ((status = nc3_def_dim (f, out_dims[d], dim_len[d], &dimid[d])))
         return handle_nc_error (who, "nc_def_dimid", (char *) out_dims[d], status);
   for (int v = 0; v < N_OUT_VARS; v++) {
      const out_var *o = &out_vars[v];
      int ids[2] = { dimid[o->dims[0]], dimid[o->dims[1]] }, varid;

      if ((status = nc3_def_var (f, o->name, o->type, o->ndims, ids, &varid)))
         return handle_nc_error (who, "nc_def_var", (char *) o->name, status);
      for (int a = 0; a < 3 && o->atts[a].name; a++)
         if ((status = nc3_put_att_text (f, varid, o->atts[a].name, strlen (o->atts[a].text), o->atts[a].text)))
            return handle_nc_error (who, "nc_put_att_text", (char *) o->name, status);
   }
   if ((status = nc3_close (f)))
      return handle_nc_error (who, "nc_close", fname, status);

   if (put_var_1d_double (fname, "z_t", z_t) || put_var_2d_double (fname, "TLONG", TLONG) ||
       put_var_2d_double (fname, "TLAT", TLAT) || put_var_2d_int (fname, "KMT", KMT))
      return 1;
   say ("exiting", who);
   return 0;
}

void free_grid_info (void)
{
   free (z_t);
   free (dz);
   free_2d_double (TLONG);
   free_2d_double (TLAT);
   free_2d_double (TAREA);
   free_2d_int (KMT);
   free_2d_int (KMU);
   z_t = dz = NULL;
   TLONG = TLAT = TAREA = NULL;
   KMT = KMU = NULL;
}
