/* Whole-variable NetCDF helpers with the reference's names and error behaviour
 * (reference src/file_io.c:10-368), implemented on nc3_codec because libnetcdf is absent.
 * Each call opens the file, looks the variable up, moves the whole variable and closes,
 * like the reference; failures print the reference's 5-line message and return the status. */
#include <stdio.h>

#include "nc3_codec.h"
#include "nkp_host.h"

int handle_nc_error (char *subname, char *cdf_subname, char *msg, int status)
{
   fprintf (stderr,
            "(%d) ERROR returned from netCDF routine\n\tsubname     : %s\n\tcdf_subname : %s\n"
            "\tmsg         : %s\n\tnetCDF msg  : %s\n", iam, subname, cdf_subname, msg, nc3_strerror (status));
   return status;
}

typedef enum { XFER_GET_INT, XFER_PUT_INT, XFER_GET_DOUBLE, XFER_PUT_DOUBLE } xfer_t;

/* one open / inquire / transfer / close cycle; `slab` is the first element of the caller's
 * contiguous storage (field, field[0] or field[0][0] as in reference src/file_io.c:286,361) */
static int whole_var (char *subname, char *fname, char *varname, xfer_t what, void *slab)
{
   static char *xfer_name[] = { "nc_get_var_int", "nc_put_var_int", "nc_get_var_double", "nc_put_var_double" };
   nc3_file *f;
   int status, varid;
   int writing = (what == XFER_PUT_INT || what == XFER_PUT_DOUBLE);

   if ((status = nc3_open (fname, writing, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   if ((status = nc3_inq_varid (f, varname, &varid))) {
      nc3_close (f);
      return handle_nc_error (subname, "nc_inq_varid", varname, status);
   }
   switch (what) {
   case XFER_GET_INT: status = nc3_get_var_int (f, varid, (int *) slab); break;
   case XFER_PUT_INT: status = nc3_put_var_int (f, varid, (const int *) slab); break;
   case XFER_GET_DOUBLE: status = nc3_get_var_double (f, varid, (double *) slab); break;
   case XFER_PUT_DOUBLE: status = nc3_put_var_double (f, varid, (const double *) slab); break;
   }
   if (status) {
      nc3_close (f);
      return handle_nc_error (subname, xfer_name[what], varname, status);
   }
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);
   return 0;
}

int get_var_1d_int (char *fname, char *varname, int *field) { return whole_var ("get_var_1d_int", fname, varname, XFER_GET_INT, field); }
int get_var_2d_int (char *fname, char *varname, int **field) { return whole_var ("get_var_2d_int", fname, varname, XFER_GET_INT, field[0]); }
int get_var_3d_int (char *fname, char *varname, int ***field) { return whole_var ("get_var_3d_int", fname, varname, XFER_GET_INT, field[0][0]); }
int put_var_1d_int (char *fname, char *varname, int *field) { return whole_var ("put_var_1d_int", fname, varname, XFER_PUT_INT, field); }
int put_var_2d_int (char *fname, char *varname, int **field) { return whole_var ("put_var_2d_int", fname, varname, XFER_PUT_INT, field[0]); }
int put_var_3d_int (char *fname, char *varname, int ***field) { return whole_var ("put_var_3d_int", fname, varname, XFER_PUT_INT, field[0][0]); }
int get_var_1d_double (char *fname, char *varname, double *field) { return whole_var ("get_var_1d_double", fname, varname, XFER_GET_DOUBLE, field); }
int get_var_2d_double (char *fname, char *varname, double **field) { return whole_var ("get_var_2d_double", fname, varname, XFER_GET_DOUBLE, field[0]); }
int get_var_3d_double (char *fname, char *varname, double ***field) { return whole_var ("get_var_3d_double", fname, varname, XFER_GET_DOUBLE, field[0][0]); }
int put_var_1d_double (char *fname, char *varname, double *field) { return whole_var ("put_var_1d_double", fname, varname, XFER_PUT_DOUBLE, field); }
int put_var_2d_double (char *fname, char *varname, double **field) { return whole_var ("put_var_2d_double", fname, varname, XFER_PUT_DOUBLE, field[0]); }
int put_var_3d_double (char *fname, char *varname, double ***field) { return whole_var ("put_var_3d_double", fname, varname, XFER_PUT_DOUBLE, field[0][0]); }

/* elements [first, first + count) of a 1-D variable: what a rank of solve_ABdist reads of the matrix arrays instead of the
 * whole variable (the reference has rank 0 read everything and send slices, src/solve_ABdist.c:141-225) */
static int part_var (char *subname, char *fname, char *varname, int as_double, size_t first, size_t count, void *out)
{
   nc3_file *f;
   int status, varid;
   if ((status = nc3_open (fname, 0, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   if ((status = nc3_inq_varid (f, varname, &varid))) {
      nc3_close (f);
      return handle_nc_error (subname, "nc_inq_varid", varname, status);
   }
   status = as_double ? nc3_get_vara_double (f, varid, first, count, (double *) out) : nc3_get_vara_int (f, varid, first, count, (int *) out);
   if (status) {
      nc3_close (f);
      return handle_nc_error (subname, as_double ? "nc_get_vara_double" : "nc_get_vara_int", varname, status);
   }
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);
   return 0;
}

int get_vara_1d_int (char *fname, char *varname, size_t first, size_t count, int *field) { return part_var ("get_vara_1d_int", fname, varname, 0, first, count, field); }
int get_vara_1d_double (char *fname, char *varname, size_t first, size_t count, double *field) { return part_var ("get_vara_1d_double", fname, varname, 1, first, count, field); }

int var_exists_in_file (char *fname, char *varname, int *retval)
{
   char *subname = "var_exists_in_file";
   nc3_file *f;
   int status, varid;

   if ((status = nc3_open (fname, 0, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   status = nc3_inq_varid (f, varname, &varid);
   *retval = (status == NC3_NOERR);           /* "not found" is an answer, not an error */
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);
   return 0;
}

int get_att_double (char *fname, char *varname, char *attname, double *val)
{
   char *subname = "get_att_double";
   nc3_file *f;
   int status, varid;

   if ((status = nc3_open (fname, 0, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   if ((status = nc3_inq_varid (f, varname, &varid))) {
      nc3_close (f);
      return handle_nc_error (subname, "nc_inq_varid", varname, status);
   }
   if ((status = nc3_get_att_double (f, varid, attname, val))) {
      nc3_close (f);
      return handle_nc_error (subname, "nc_get_att_double", varname, status);
   }
   if ((status = nc3_close (f)))
      return handle_nc_error (subname, "nc_close", fname, status);
   return 0;
}

/* addition of this build: total element count of a variable, so callers can check it against
 * km*jmt*imt BEFORE a whole-variable read lands in their buffer (the reference trusts the file) */
int nkp_var_nelems (char *fname, char *varname, size_t *nelems)
{
   char *subname = "nkp_var_nelems";
   nc3_file *f;
   int status, varid;

   if ((status = nc3_open (fname, 0, &f)))
      return handle_nc_error (subname, "nc_open", fname, status);
   if ((status = nc3_inq_varid (f, varname, &varid))) {
      nc3_close (f);
      return handle_nc_error (subname, "nc_inq_varid", varname, status);
   }
   status = nc3_inq_var (f, varid, NULL, NULL, nelems);
   nc3_close (f);
   return status ? handle_nc_error (subname, "nc_inq_var", varname, status) : 0;
}
