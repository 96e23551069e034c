/* NetCDF-4 / HDF5 files for the whole-variable surface of nc3_codec.h, through a libhdf5 found at run time (nc4_hdf5.c).
 * Internal to libnkp_host: nc3_open hands an HDF5 container to nc4_open and the nc3_* calls dispatch here. */
#ifndef NC4_HDF5_H
#define NC4_HDF5_H
#include <stddef.h>
#include <stdint.h>

typedef struct nc4_file nc4_file;

int nc4_available (void);                                    /* 1 when a usable libhdf5 (>= 1.10) could be loaded */
int nc4_open (const char *path, int writable, nc4_file **out);   /* NC3_EHDF5 when no libhdf5 is available */
int nc4_close (nc4_file *f);
int nc4_inq_dimlen (nc4_file *f, const char *name, size_t *len);
int nc4_inq_varid (nc4_file *f, const char *name, int *varid);
int nc4_inq_var (nc4_file *f, int varid, int *nc_type, int *ndims, size_t *nelems, size_t *dimlens /* ndims entries, may be NULL */);
/* whole variable (count = UINT64_MAX) or elements [first, first + count) of a 1-D variable; memory type double or int */
int nc4_transfer (nc4_file *f, int varid, int as_double, int writing, uint64_t first, uint64_t count, void *mem);
int nc4_get_att_double (nc4_file *f, int varid, const char *attname, double *val);
#endif
