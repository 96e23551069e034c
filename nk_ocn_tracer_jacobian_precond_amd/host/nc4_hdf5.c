/* NetCDF-4 (HDF5 container) files behind the same surface as the classic codec: see nc4_hdf5.h.
 *
 * CESM history files -- the inputs of the reference's gen_A and the tracer files of its solvers
 * (reference test/test_gen_A.csh:13-15, test/test_solve_ABglobal.csh:21-26) -- are commonly netCDF-4.
 * The reference reads them through libnetcdf (src/file_io.c:3); this image has neither libnetcdf nor its
 * headers, but a libhdf5 may be present.  It is loaded at run time (dlopen: no link-time dependency; a
 * machine without it keeps the clear refusal of the classic codec) and only the handful of calls below is
 * used.  What the netCDF-4 format guarantees and this file relies on:
 *   - every variable is an HDF5 dataset of the same name in the root group, its shape the variable's shape;
 *   - every dimension has a dataset of its name as well: its coordinate variable, or -- for a dimension
 *     without one -- a placeholder dataset (NAME attribute "This is a netCDF dimension but not a netCDF
 *     variable") whose extent is the dimension's current length;
 *   - _FillValue is an attribute of the dataset's own type.
 * Type conversion (float / short / int64 -> double, etc.), chunking and compression filters are libhdf5's.
 */
#define _GNU_SOURCE
#include "nc4_hdf5.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nc3_codec.h"

typedef int64_t hid_t;               /* HDF5 >= 1.10; older libraries (32-bit ids) are refused */
typedef int herr_t;
typedef int htri_t;
typedef unsigned long long hsize_t;

#define H5F_ACC_RDONLY 0u
#define H5F_ACC_RDWR 1u
#define H5P_DEFAULT ((hid_t) 0)
#define H5S_ALL ((hid_t) 0)
#define H5E_DEFAULT ((hid_t) 0)
#define H5T_INTEGER 0
#define H5T_FLOAT 1
#define H5S_SELECT_SET 0
#define H5T_SGN_NONE 0

static struct {
   void *lib;
   int tried, ok;
   herr_t (*H5open) (void);
   herr_t (*H5get_libversion) (unsigned *, unsigned *, unsigned *);
   herr_t (*H5Eset_auto2) (hid_t, void *, void *);
   hid_t (*H5Fopen) (const char *, unsigned, hid_t);
   herr_t (*H5Fclose) (hid_t);
   htri_t (*H5Lexists) (hid_t, const char *, hid_t);
   hid_t (*H5Dopen2) (hid_t, const char *, hid_t);
   herr_t (*H5Dclose) (hid_t);
   hid_t (*H5Dget_space) (hid_t);
   hid_t (*H5Dget_type) (hid_t);
   herr_t (*H5Dread) (hid_t, hid_t, hid_t, hid_t, hid_t, void *);
   herr_t (*H5Dwrite) (hid_t, hid_t, hid_t, hid_t, hid_t, const void *);
   int (*H5Sget_simple_extent_ndims) (hid_t);
   int (*H5Sget_simple_extent_dims) (hid_t, hsize_t *, hsize_t *);
   hid_t (*H5Screate_simple) (int, const hsize_t *, const hsize_t *);
   herr_t (*H5Sselect_hyperslab) (hid_t, int, const hsize_t *, const hsize_t *, const hsize_t *, const hsize_t *);
   herr_t (*H5Sclose) (hid_t);
   int (*H5Tget_class) (hid_t);
   size_t (*H5Tget_size) (hid_t);
   int (*H5Tget_sign) (hid_t);
   herr_t (*H5Tclose) (hid_t);
   htri_t (*H5Aexists) (hid_t, const char *);
   hid_t (*H5Aopen) (hid_t, const char *, hid_t);
   herr_t (*H5Aread) (hid_t, hid_t, void *);
   herr_t (*H5Aclose) (hid_t);
   hid_t native_double, native_int;
} H;

static int load_hdf5 (void)
{
   if (H.tried) return H.ok;
   H.tried = 1;
   const char *want = getenv ("NKP_HDF5_LIB");                      /* a path, or "none" to keep netCDF-4 support off */
   if (want && !strcmp (want, "none")) return 0;
   const char *names[] = { want, "libhdf5.so", "libhdf5.so.310", "libhdf5.so.200", "libhdf5.so.103", "libhdf5_serial.so",
                           "/opt/conda/lib/libhdf5.so" };
   for (size_t i = 0; i < sizeof names / sizeof names[0] && !H.lib; i++)
      if (names[i] && *names[i]) H.lib = dlopen (names[i], RTLD_NOW | RTLD_LOCAL);
   if (!H.lib) return 0;
#define SYM(name) do { *(void **) (&H.name) = dlsym (H.lib, #name); if (!H.name) return 0; } while (0)
   SYM (H5open); SYM (H5get_libversion); SYM (H5Eset_auto2); SYM (H5Fopen); SYM (H5Fclose); SYM (H5Lexists); SYM (H5Dopen2); SYM (H5Dclose);
   SYM (H5Dget_space); SYM (H5Dget_type); SYM (H5Dread); SYM (H5Dwrite); SYM (H5Sget_simple_extent_ndims); SYM (H5Sget_simple_extent_dims);
   SYM (H5Screate_simple); SYM (H5Sselect_hyperslab); SYM (H5Sclose); SYM (H5Tget_class); SYM (H5Tget_size); SYM (H5Tget_sign); SYM (H5Tclose);
   SYM (H5Aexists); SYM (H5Aopen); SYM (H5Aread); SYM (H5Aclose);
#undef SYM
   unsigned maj = 0, min = 0, rel = 0;
   if (H.H5open () < 0 || H.H5get_libversion (&maj, &min, &rel) < 0) return 0;
   if (maj < 1 || (maj == 1 && min < 10)) return 0;                 /* 32-bit hid_t: not the ABI declared above */
   hid_t *pd = (hid_t *) dlsym (H.lib, "H5T_NATIVE_DOUBLE_g"), *pi = (hid_t *) dlsym (H.lib, "H5T_NATIVE_INT_g");
   if (!pd || !pi) return 0;
   H.native_double = *pd;
   H.native_int = *pi;
   (void) H.H5Eset_auto2 (H5E_DEFAULT, NULL, NULL);                 /* errors come back as status codes, not as a printed stack */
   H.ok = 1;
   return 1;
}

#define NC4_MAX_VARS 4096
struct nc4_file {
   hid_t file;
   int writable;
   int nvars;
   hid_t dset[NC4_MAX_VARS];         /* datasets opened so far; the index is the variable id */
   char *name[NC4_MAX_VARS];
};

int nc4_available (void) { return load_hdf5 (); }

int nc4_open (const char *path, int writable, nc4_file **out)
{
   *out = NULL;
   if (!load_hdf5 ()) return NC3_EHDF5;
   hid_t fid = H.H5Fopen (path, writable ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT);
   if (fid < 0) {
      /* a sound file that only lacks write permission is NC3_EPERM, anything else is not a usable HDF5 file */
      hid_t probe = writable ? H.H5Fopen (path, H5F_ACC_RDONLY, H5P_DEFAULT) : (hid_t) -1;
      if (probe >= 0) { (void) H.H5Fclose (probe); return NC3_EPERM; }
      return NC3_EHDF5OPEN;
   }
   nc4_file *f = (nc4_file *) calloc (1, sizeof (nc4_file));
   if (!f) { (void) H.H5Fclose (fid); return NC3_ENOMEM; }
   f->file = fid;
   f->writable = writable;
   *out = f;
   return NC3_NOERR;
}

int nc4_close (nc4_file *f)
{
   if (!f) return NC3_NOERR;
   int status = NC3_NOERR;
   for (int i = 0; i < f->nvars; i++) {
      if (H.H5Dclose (f->dset[i]) < 0) status = NC3_EIO;
      free (f->name[i]);
   }
   if (H.H5Fclose (f->file) < 0) status = NC3_EIO;
   free (f);
   return status;
}

static int open_dataset (nc4_file *f, const char *name, int *id)
{
   for (int i = 0; i < f->nvars; i++)
      if (!strcmp (f->name[i], name)) { *id = i; return NC3_NOERR; }
   if (H.H5Lexists (f->file, name, H5P_DEFAULT) <= 0) return NC3_ENOTVAR;
   if (f->nvars >= NC4_MAX_VARS) return NC3_ENOMEM;
   hid_t d = H.H5Dopen2 (f->file, name, H5P_DEFAULT);
   if (d < 0) return NC3_ENOTVAR;                     /* a group of that name is not a variable */
   f->dset[f->nvars] = d;
   f->name[f->nvars] = strdup (name);
   *id = f->nvars++;
   return NC3_NOERR;
}

static int extent (hid_t dset, int *ndims, hsize_t *dims /* 32 */, uint64_t *nelems)
{
   hid_t sp = H.H5Dget_space (dset);
   if (sp < 0) return NC3_EIO;
   int nd = H.H5Sget_simple_extent_ndims (sp);
   if (nd < 0 || nd > 32 || (nd > 0 && H.H5Sget_simple_extent_dims (sp, dims, NULL) < 0)) { (void) H.H5Sclose (sp); return NC3_EIO; }
   (void) H.H5Sclose (sp);
   uint64_t n = 1;
   for (int d = 0; d < nd; d++) n *= (uint64_t) dims[d];
   *ndims = nd;
   *nelems = n;
   return NC3_NOERR;
}

int nc4_inq_dimlen (nc4_file *f, const char *name, size_t *len)
{
   int id, nd;
   hsize_t dims[32];
   uint64_t n;
   if (open_dataset (f, name, &id)) return NC3_EBADDIM;
   if (extent (f->dset[id], &nd, dims, &n)) return NC3_EIO;
   if (nd != 1) return NC3_EBADDIM;                   /* a dimension's dataset is one-dimensional */
   *len = (size_t) dims[0];
   return NC3_NOERR;
}

int nc4_inq_varid (nc4_file *f, const char *name, int *varid) { return open_dataset (f, name, varid); }

int nc4_inq_var (nc4_file *f, int varid, int *nc_type, int *ndims, size_t *nelems, size_t *dimlens)
{
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   int nd;
   hsize_t dims[32];
   uint64_t n;
   int status = extent (f->dset[varid], &nd, dims, &n);
   if (status) return status;
   if (nc_type) {
      hid_t t = H.H5Dget_type (f->dset[varid]);
      if (t < 0) return NC3_EIO;
      const int cls = H.H5Tget_class (t);
      const size_t sz = H.H5Tget_size (t);
      const int uns = cls == H5T_INTEGER && H.H5Tget_sign (t) == H5T_SGN_NONE;
      (void) H.H5Tclose (t);
      if (cls == H5T_FLOAT) *nc_type = sz == 4 ? NC3_FLOAT : NC3_DOUBLE;
      else if (cls == H5T_INTEGER) *nc_type = sz == 1 ? (uns ? NC3_UBYTE : NC3_BYTE) : sz == 2 ? (uns ? NC3_USHORT : NC3_SHORT) : sz == 4 ? (uns ? NC3_UINT : NC3_INT) : (uns ? NC3_UINT64 : NC3_INT64);
      else return NC3_ENOTNC;                         /* strings, compounds: not a numeric variable */
   }
   if (ndims) *ndims = nd;
   if (nelems) *nelems = (size_t) n;
   if (dimlens)
      for (int d = 0; d < nd; d++) dimlens[d] = (size_t) dims[d];
   return NC3_NOERR;
}

int nc4_transfer (nc4_file *f, int varid, int as_double, int writing, uint64_t first, uint64_t count, void *mem)
{
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   if (writing && !f->writable) return NC3_EPERM;
   const hid_t mt = as_double ? H.native_double : H.native_int;
   hid_t msp = H5S_ALL, fsp = H5S_ALL;
   if (count != UINT64_MAX) {
      /* a range of a one-dimensional variable */
      int nd;
      hsize_t dims[32];
      uint64_t n;
      if (extent (f->dset[varid], &nd, dims, &n)) return NC3_EIO;
      if (nd != 1 || first > n || count > n - first) return NC3_EEDGE;
      const hsize_t start = (hsize_t) first, cnt = (hsize_t) count;
      fsp = H.H5Dget_space (f->dset[varid]);
      msp = H.H5Screate_simple (1, &cnt, NULL);
      if (fsp < 0 || msp < 0 || (count && H.H5Sselect_hyperslab (fsp, H5S_SELECT_SET, &start, NULL, &cnt, NULL) < 0)) {
         if (fsp >= 0) (void) H.H5Sclose (fsp);
         if (msp >= 0) (void) H.H5Sclose (msp);
         return NC3_EIO;
      }
      if (count == 0) { (void) H.H5Sclose (fsp); (void) H.H5Sclose (msp); return NC3_NOERR; }
   }
   const herr_t e = writing ? H.H5Dwrite (f->dset[varid], mt, msp, fsp, H5P_DEFAULT, mem) : H.H5Dread (f->dset[varid], mt, msp, fsp, H5P_DEFAULT, mem);
   if (fsp != H5S_ALL) (void) H.H5Sclose (fsp);
   if (msp != H5S_ALL) (void) H.H5Sclose (msp);
   return e < 0 ? NC3_EIO : NC3_NOERR;
}

int nc4_get_att_double (nc4_file *f, int varid, const char *attname, double *val)
{
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   if (H.H5Aexists (f->dset[varid], attname) <= 0) return NC3_ENOTATT;
   hid_t a = H.H5Aopen (f->dset[varid], attname, H5P_DEFAULT);
   if (a < 0) return NC3_ENOTATT;
   const herr_t e = H.H5Aread (a, H.native_double, val);
   (void) H.H5Aclose (a);
   return e < 0 ? NC3_EIO : NC3_NOERR;
}
