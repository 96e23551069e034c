/* NetCDF classic-format codec (CDF-1, CDF-2 = 64-bit offset, CDF-5 = 64-bit data).
 *
 * The reference links libnetcdf (reference src/file_io.c:3, src/Makefile:12-13); this image
 * has none, and the hot path only ever does "open, look a variable up, read or write ALL of
 * it, close" (reference src/file_io.c:72-93, 222-243, 272-293, 347-368).  This codec is
 * exactly that much of the format, written from the published classic-format grammar.
 *
 * Error codes follow libnetcdf's numbering for the cases the reference can hit so messages
 * printed through handle_nc_error stay recognisable.
 */
#ifndef NKP_NC3_CODEC_H
#define NKP_NC3_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
   NC3_NOERR = 0,
   NC3_EBADDIM = -46,     /* dimension not found      */
   NC3_ENOTVAR = -49,     /* variable not found       */
   NC3_ENOTNC = -51,      /* not a classic netCDF file */
   NC3_ERANGE = -60,      /* value not representable in the target type */
   NC3_ENOMEM = -61,
   NC3_EHDF5 = -101,      /* netCDF-4/HDF5 container and no libhdf5 could be loaded to read it */
   NC3_EHDF5OPEN = -102,  /* HDF5 signature, but libhdf5 cannot open the file (damaged, or not a netCDF-4 / HDF5 file at all) */
   NC3_ENOTATT = -43,
   NC3_EIO = -68,         /* open/seek/short read/short write */
   NC3_EPERM = -37,       /* write to a file opened read-only */
   NC3_ENAMEINUSE = -42,  /* dimension / variable name already defined */
   NC3_ENOTINDEFINE = -38,/* definition call outside define mode */
   NC3_EINDEFINE = -39,   /* data call while in define mode */
   NC3_EINVAL = -36,
   NC3_EEDGE = -57        /* start + count exceeds the variable (libnetcdf's NC_EEDGE), or a range of a record variable */
};

enum { NC3_BYTE = 1, NC3_CHAR, NC3_SHORT, NC3_INT, NC3_FLOAT, NC3_DOUBLE,
       NC3_UBYTE, NC3_USHORT, NC3_UINT, NC3_INT64, NC3_UINT64 };

typedef struct nc3_file nc3_file;

/* mode: 0 = read-only (NC_NOWRITE), 1 = read-write (NC_WRITE) */
int nc3_open (const char *path, int writable, nc3_file **out);
int nc3_close (nc3_file *f);
const char *nc3_strerror (int status);

int nc3_inq_dimlen (nc3_file *f, const char *dimname, size_t *len);
int nc3_inq_varid (nc3_file *f, const char *varname, int *varid);
/* total element count of a variable (records included) and its external type */
int nc3_inq_var (nc3_file *f, int varid, int *nc_type, int *ndims, size_t *nelems);
int nc3_inq_var_dimlens (nc3_file *f, int varid, size_t *dimlens /* ndims entries */);

/* whole-variable transfers with libnetcdf-style type conversion */
int nc3_get_var_double (nc3_file *f, int varid, double *out);
int nc3_get_var_int (nc3_file *f, int varid, int *out);
int nc3_put_var_double (nc3_file *f, int varid, const double *in);
/* elements [first, first + count) of a fixed-size variable in storage order (nc_get_vara on a 1-D variable) */
int nc3_get_vara_double (nc3_file *f, int varid, size_t first, size_t count, double *out);
int nc3_get_vara_int (nc3_file *f, int varid, size_t first, size_t count, int *out);
int nc3_put_var_int (nc3_file *f, int varid, const int *in);
/* numeric attribute of a variable (varid -1 = global), first element, as double */
int nc3_get_att_double (nc3_file *f, int varid, const char *attname, double *val);

/* ---- define mode: the calls the reference's writers make (reference src/grid.c:234-296,
 * src/matrix.c:283-333, 3862-3895): create a 64-bit-offset file or re-open one with nc_redef,
 * add dimensions, variables and attributes, leave define mode (nc_close does it implicitly,
 * as in libnetcdf).  Leaving define mode lays the variables out in definition order and moves
 * existing data behind the grown header; new variables are pre-filled with the default fill
 * values (libnetcdf's NC_FILL default).  New record variables are not supported. */
int nc3_create (const char *path, int version /* 1, 2 or 5 */, nc3_file **out);
int nc3_redef (nc3_file *f);
/* fill != 0 (default): pre-fill new variables at nc3_enddef (NC_FILL); 0: leave them unwritten (NC_NOFILL) */
int nc3_set_fill (nc3_file *f, int fill);
int nc3_enddef (nc3_file *f);
int nc3_def_dim (nc3_file *f, const char *name, size_t len, int *dimid);
int nc3_inq_dimid (nc3_file *f, const char *name, int *dimid);
int nc3_def_var (nc3_file *f, const char *name, int nc_type, int ndims, const int *dimids, int *varid);
int nc3_put_att_text (nc3_file *f, int varid, const char *name, size_t len, const char *text);
int nc3_put_att_int (nc3_file *f, int varid, const char *name, int nc_type, size_t n, const int *vals);
int nc3_put_att_double (nc3_file *f, int varid, const char *name, int nc_type, size_t n, const double *vals);

#ifdef __cplusplus
}
#endif
#endif
