/* NetCDF classic-format codec: see nc3_codec.h.  Written from the classic-format grammar
 * (header = magic numrecs dim_list gatt_list var_list; big-endian; names and attribute
 * payloads padded to 4 bytes; record variables interleaved per record). */
#define _FILE_OFFSET_BITS 64
#include "nc3_codec.h"
#include "nc4_hdf5.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define TAG_DIM 0x0A
#define TAG_VAR 0x0B
#define TAG_ATT 0x0C
#define CHUNK_ELEMS ((size_t) 1 << 20)

typedef struct { char *name; uint64_t len; } dim_t;
typedef struct { char *name; int type; uint64_t n; unsigned char *raw; } att_t;
typedef struct {
   char *name;
   int ndims;
   int *dimids;
   int natts;
   att_t *atts;
   int type;
   uint64_t vsize;
   int64_t begin;
   int is_record;
   uint64_t per_rec;            /* elements per record (or in total when not a record var) */
} var_t;

struct nc3_file {
   nc4_file *h4;                /* a netCDF-4 / HDF5 container: every call goes to nc4_hdf5.c, nothing below is used */
   FILE *fp;
   char *path;
   int writable;
   int defining;                /* between nc3_create / nc3_redef and nc3_enddef */
   int nofill;                  /* NC_NOFILL: new variables are not pre-filled */
   int nvars_on_disk;           /* variables that already have data in the file */
   int64_t *old_begin;          /* their offsets before the header grew */
   int version;                 /* 1, 2, 5 */
   uint64_t numrecs;
   int ndims, nvars, ngatts;
   dim_t *dims;
   var_t *vars;
   att_t *gatts;
   uint64_t recsize;
};

static const int type_size[12] = { 0, 1, 1, 2, 4, 4, 8, 1, 2, 4, 8, 8 };

/* ---------------------------------------------------------------- header reader */

typedef struct { FILE *fp; int ok; int wide; } rd_t;

static uint32_t rd_u32 (rd_t *r)
{
   unsigned char b[4];
   if (fread (b, 1, 4, r->fp) != 4) { r->ok = 0; return 0; }
   return ((uint32_t) b[0] << 24) | ((uint32_t) b[1] << 16) | ((uint32_t) b[2] << 8) | b[3];
}

static uint64_t rd_u64 (rd_t *r)
{
   uint64_t hi = rd_u32 (r);
   uint64_t lo = rd_u32 (r);
   return (hi << 32) | lo;
}

static uint64_t rd_nonneg (rd_t *r) { return r->wide ? rd_u64 (r) : rd_u32 (r); }

static char *rd_name (rd_t *r)
{
   uint64_t n = rd_nonneg (r);
   if (!r->ok || n > (1u << 20)) { r->ok = 0; return NULL; }
   size_t padded = (size_t) ((n + 3) & ~(uint64_t) 3);
   char *s = (char *) malloc (padded + 1);
   if (!s) { r->ok = 0; return NULL; }
   if (padded && fread (s, 1, padded, r->fp) != padded) r->ok = 0;
   s[n] = '\0';
   return s;
}

static int rd_attlist (rd_t *r, int *natts, att_t **atts)
{
   uint32_t tag = rd_u32 (r);
   uint64_t cnt = rd_nonneg (r);
   *natts = 0;
   *atts = NULL;
   if (!r->ok) return NC3_ENOTNC;
   if (tag == 0 && cnt == 0) return NC3_NOERR;
   if (tag != TAG_ATT || cnt > (1u << 20)) return NC3_ENOTNC;
   att_t *a = (att_t *) calloc ((size_t) cnt ? (size_t) cnt : 1, sizeof (att_t));
   if (!a) return NC3_ENOMEM;
   for (uint64_t i = 0; i < cnt; i++) {
      a[i].name = rd_name (r);
      a[i].type = (int) rd_u32 (r);
      a[i].n = rd_nonneg (r);
      if (!r->ok || a[i].type < 1 || a[i].type > 11) { *natts = (int) i; *atts = a; return NC3_ENOTNC; }
      uint64_t nbytes = a[i].n * (uint64_t) type_size[a[i].type];
      size_t padded = (size_t) ((nbytes + 3) & ~(uint64_t) 3);
      a[i].raw = (unsigned char *) malloc (padded ? padded : 1);
      if (!a[i].raw) { *natts = (int) i; *atts = a; return NC3_ENOMEM; }
      if (padded && fread (a[i].raw, 1, padded, r->fp) != padded) r->ok = 0;
   }
   *natts = (int) cnt;
   *atts = a;
   return r->ok ? NC3_NOERR : NC3_ENOTNC;
}

static void free_atts (int n, att_t *a)
{
   if (!a) return;
   for (int i = 0; i < n; i++) { free (a[i].name); free (a[i].raw); }
   free (a);
}

int nc3_close (nc3_file *f)
{
   int status = NC3_NOERR;
   if (!f) return NC3_NOERR;
   if (f->h4) { status = nc4_close (f->h4); free (f); return status; }
   if (f->defining) status = nc3_enddef (f);
   if (f->fp && fclose (f->fp) && !status) status = NC3_EIO;
   for (int i = 0; i < f->ndims; i++) free (f->dims[i].name);
   free (f->dims);
   for (int i = 0; i < f->nvars; i++) {
      free (f->vars[i].name);
      free (f->vars[i].dimids);
      free_atts (f->vars[i].natts, f->vars[i].atts);
   }
   free (f->vars);
   free_atts (f->ngatts, f->gatts);
   free (f->path);
   free (f->old_begin);
   free (f);
   return status;
}

int nc3_open (const char *path, int writable, nc3_file **out)
{
   *out = NULL;
   FILE *fp = fopen (path, writable ? "r+b" : "rb");
   if (!fp) return NC3_EIO;
   unsigned char magic[4];
   if (fread (magic, 1, 4, fp) != 4) { fclose (fp); return NC3_ENOTNC; }
   if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') {
      /* netCDF-4: read and written through libhdf5 when one can be loaded, refused (NC3_EHDF5) otherwise */
      fclose (fp);
      nc4_file *h4 = NULL;
      const int st = nc4_open (path, writable, &h4);
      if (st) return st;
      nc3_file *f4 = (nc3_file *) calloc (1, sizeof (nc3_file));
      if (!f4) { nc4_close (h4); return NC3_ENOMEM; }
      f4->h4 = h4;
      f4->writable = writable;
      *out = f4;
      return NC3_NOERR;
   }
   if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2 && magic[3] != 5)) {
      fclose (fp);
      return NC3_ENOTNC;
   }
   nc3_file *f = (nc3_file *) calloc (1, sizeof (nc3_file));
   if (!f) { fclose (fp); return NC3_ENOMEM; }
   f->fp = fp;
   f->path = strdup (path);
   f->writable = writable;
   f->version = magic[3];
   rd_t r = { fp, 1, f->version == 5 };
   int status = NC3_NOERR;

   f->numrecs = rd_nonneg (&r);
   if (!r.wide && f->numrecs == 0xFFFFFFFFu) f->numrecs = 0;      /* STREAMING marker */

   uint32_t tag = rd_u32 (&r);
   uint64_t cnt = rd_nonneg (&r);
   if (!r.ok || (tag != 0 && tag != TAG_DIM) || cnt > (1u << 20)) { status = NC3_ENOTNC; goto fail; }
   if (tag == TAG_DIM) {
      f->dims = (dim_t *) calloc ((size_t) cnt ? (size_t) cnt : 1, sizeof (dim_t));
      if (!f->dims) { status = NC3_ENOMEM; goto fail; }
      for (uint64_t i = 0; i < cnt; i++) {
         f->dims[i].name = rd_name (&r);
         f->dims[i].len = rd_nonneg (&r);
         f->ndims = (int) i + 1;
      }
   }
   if ((status = rd_attlist (&r, &f->ngatts, &f->gatts))) goto fail;

   tag = rd_u32 (&r);
   cnt = rd_nonneg (&r);
   if (!r.ok || (tag != 0 && tag != TAG_VAR) || cnt > (1u << 20)) { status = NC3_ENOTNC; goto fail; }
   if (tag == TAG_VAR) {
      f->vars = (var_t *) calloc ((size_t) cnt ? (size_t) cnt : 1, sizeof (var_t));
      if (!f->vars) { status = NC3_ENOMEM; goto fail; }
      for (uint64_t i = 0; i < cnt; i++) {
         var_t *v = &f->vars[i];
         f->nvars = (int) i + 1;
         v->name = rd_name (&r);
         uint64_t nd = rd_nonneg (&r);
         if (!r.ok || nd > 1024) { status = NC3_ENOTNC; goto fail; }
         v->ndims = (int) nd;
         v->dimids = (int *) calloc (nd ? (size_t) nd : 1, sizeof (int));
         if (!v->dimids) { status = NC3_ENOMEM; goto fail; }
         for (uint64_t d = 0; d < nd; d++) {
            uint64_t id = rd_nonneg (&r);
            if (id >= (uint64_t) f->ndims) { status = NC3_ENOTNC; goto fail; }
            v->dimids[d] = (int) id;
         }
         if ((status = rd_attlist (&r, &v->natts, &v->atts))) goto fail;
         v->type = (int) rd_u32 (&r);
         v->vsize = rd_nonneg (&r);
         v->begin = (f->version == 1) ? (int64_t) rd_u32 (&r) : (int64_t) rd_u64 (&r);
         if (!r.ok || v->type < 1 || v->type > 11) { status = NC3_ENOTNC; goto fail; }
         v->is_record = (v->ndims > 0 && f->dims[v->dimids[0]].len == 0);
         v->per_rec = 1;
         for (int d = v->is_record ? 1 : 0; d < v->ndims; d++) v->per_rec *= f->dims[v->dimids[d]].len;
      }
   }
   if (!r.ok || !f->path) { status = r.ok ? NC3_ENOMEM : NC3_ENOTNC; goto fail; }
   f->nvars_on_disk = f->nvars;

   {  /* record stride: sum of the record variables' (padded) vsize, unpadded if there is only one */
      int nrec = 0;
      uint64_t sum = 0, single = 0;
      for (int i = 0; i < f->nvars; i++)
         if (f->vars[i].is_record) {
            nrec++;
            uint64_t raw = f->vars[i].per_rec * (uint64_t) type_size[f->vars[i].type];
            sum += (raw + 3) & ~(uint64_t) 3;
            single = raw;
         }
      f->recsize = (nrec == 1) ? single : sum;
   }
   *out = f;
   return NC3_NOERR;

 fail:
   nc3_close (f);
   return status;
}

const char *nc3_strerror (int status)
{
   switch (status) {
   case NC3_NOERR: return "No error";
   case NC3_EBADDIM: return "NetCDF: Invalid dimension ID or name";
   case NC3_ENOTVAR: return "NetCDF: Variable not found";
   case NC3_ENOTNC: return "NetCDF: Unknown file format";
   case NC3_ERANGE: return "NetCDF: Numeric conversion not representable";
   case NC3_ENOMEM: return "NetCDF: Memory allocation (malloc) failure";
   case NC3_EHDF5OPEN: return "NetCDF: the file carries the HDF5 signature but libhdf5 could not open it";
   case NC3_EHDF5: return "NetCDF: a netCDF-4/HDF5 file, and no libhdf5 (>= 1.10) could be loaded to read it (NKP_HDF5_LIB names one; classic CDF-1/2/5 files need none; nccopy -k cdf5 converts)";
   case NC3_ENOTATT: return "NetCDF: Attribute not found";
   case NC3_EIO: return "NetCDF: I/O failure (open, seek, read or write)";
   case NC3_EPERM: return "NetCDF: Write to read only";
   case NC3_ENAMEINUSE: return "NetCDF: String match to name in use";
   case NC3_ENOTINDEFINE: return "NetCDF: Operation not allowed in data mode";
   case NC3_EINDEFINE: return "NetCDF: Operation not allowed in define mode";
   case NC3_EINVAL: return "NetCDF: Invalid Argument";
   case NC3_EEDGE: return "NetCDF: Start+count exceeds dimension bound";
   default: return "NetCDF: Unknown error";
   }
}

/* ---------------------------------------------------------------- inquiries */

int nc3_inq_dimlen (nc3_file *f, const char *dimname, size_t *len)
{
   if (f->h4) return nc4_inq_dimlen (f->h4, dimname, len);
   for (int i = 0; i < f->ndims; i++)
      if (strcmp (f->dims[i].name, dimname) == 0) {
         *len = (size_t) (f->dims[i].len ? f->dims[i].len : f->numrecs);
         return NC3_NOERR;
      }
   return NC3_EBADDIM;
}

int nc3_inq_varid (nc3_file *f, const char *varname, int *varid)
{
   if (f->h4) return nc4_inq_varid (f->h4, varname, varid);
   for (int i = 0; i < f->nvars; i++)
      if (strcmp (f->vars[i].name, varname) == 0) { *varid = i; return NC3_NOERR; }
   return NC3_ENOTVAR;
}

int nc3_inq_var (nc3_file *f, int varid, int *nc_type, int *ndims, size_t *nelems)
{
   if (f->h4) return nc4_inq_var (f->h4, varid, nc_type, ndims, nelems, NULL);
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   var_t *v = &f->vars[varid];
   if (nc_type) *nc_type = v->type;
   if (ndims) *ndims = v->ndims;
   if (nelems) *nelems = (size_t) (v->per_rec * (v->is_record ? f->numrecs : 1));
   return NC3_NOERR;
}

int nc3_inq_var_dimlens (nc3_file *f, int varid, size_t *dimlens)
{
   if (f->h4) return nc4_inq_var (f->h4, varid, NULL, NULL, NULL, dimlens);
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   var_t *v = &f->vars[varid];
   for (int d = 0; d < v->ndims; d++) {
      uint64_t len = f->dims[v->dimids[d]].len;
      dimlens[d] = (size_t) ((d == 0 && v->is_record) ? f->numrecs : len);
   }
   return NC3_NOERR;
}

/* ---------------------------------------------------------------- element conversion */

static inline uint64_t load_be (const unsigned char *p, int nbytes)
{
   uint64_t u = 0;
   for (int i = 0; i < nbytes; i++) u = (u << 8) | p[i];
   return u;
}

static inline void store_be (unsigned char *p, int nbytes, uint64_t u)
{
   for (int i = nbytes - 1; i >= 0; i--) { p[i] = (unsigned char) (u & 0xFF); u >>= 8; }
}

static inline double decode_double (const unsigned char *p, int type)
{
   switch (type) {
   case NC3_BYTE: return (double) (int8_t) p[0];
   case NC3_CHAR: case NC3_UBYTE: return (double) p[0];
   case NC3_SHORT: return (double) (int16_t) load_be (p, 2);
   case NC3_USHORT: return (double) (uint16_t) load_be (p, 2);
   case NC3_INT: return (double) (int32_t) load_be (p, 4);
   case NC3_UINT: return (double) (uint32_t) load_be (p, 4);
   case NC3_FLOAT: { uint32_t u = (uint32_t) load_be (p, 4); float x; memcpy (&x, &u, 4); return (double) x; }
   case NC3_DOUBLE: { uint64_t u = load_be (p, 8); double x; memcpy (&x, &u, 8); return x; }
   case NC3_INT64: return (double) (int64_t) load_be (p, 8);
   case NC3_UINT64: return (double) load_be (p, 8);
   }
   return 0.0;
}

/* returns 1 when the value does not fit the external type */
static inline int encode_double (unsigned char *p, int type, double x)
{
   int bad = 0;
   switch (type) {
   case NC3_BYTE: bad = !(x >= -128.0 && x <= 127.0); p[0] = (unsigned char) (int8_t) (bad ? 0 : x); break;
   case NC3_CHAR: case NC3_UBYTE: bad = !(x >= 0.0 && x <= 255.0); p[0] = (unsigned char) (bad ? 0 : x); break;
   case NC3_SHORT: bad = !(x >= -32768.0 && x <= 32767.0); store_be (p, 2, (uint64_t) (uint16_t) (int16_t) (bad ? 0 : x)); break;
   case NC3_USHORT: bad = !(x >= 0.0 && x <= 65535.0); store_be (p, 2, (uint64_t) (uint16_t) (bad ? 0 : x)); break;
   case NC3_INT: bad = !(x >= -2147483648.0 && x <= 2147483647.0); store_be (p, 4, (uint64_t) (uint32_t) (int32_t) (bad ? 0 : x)); break;
   case NC3_UINT: bad = !(x >= 0.0 && x <= 4294967295.0); store_be (p, 4, (uint64_t) (uint32_t) (bad ? 0 : x)); break;
   case NC3_FLOAT: {
         float y = (float) x;
         bad = (isfinite (x) && !isfinite (y));
         uint32_t u; memcpy (&u, &y, 4); store_be (p, 4, u);
         break;
      }
   case NC3_DOUBLE: { uint64_t u; memcpy (&u, &x, 8); store_be (p, 8, u); break; }
   case NC3_INT64: bad = !(x >= -9.2233720368547758e18 && x < 9.2233720368547758e18); store_be (p, 8, (uint64_t) (int64_t) (bad ? 0 : x)); break;
   case NC3_UINT64: bad = !(x >= 0.0 && x < 1.8446744073709552e19); store_be (p, 8, (uint64_t) (bad ? 0 : x)); break;
   }
   return bad;
}

/* ---------------------------------------------------------------- whole-variable transfer */

typedef enum { AS_DOUBLE, AS_INT } mem_t;

/* first / count: a range of the variable's elements in storage order (a hyperslab of a 1-D variable); count = UINT64_MAX
 * moves the whole variable.  Ranges are for fixed-size variables: a record variable's elements are not contiguous. */
static int transfer (nc3_file *f, int varid, void *mem, mem_t mt, int writing, uint64_t first, uint64_t count)
{
   if (f->h4) return nc4_transfer (f->h4, varid, mt == AS_DOUBLE, writing, first, count, mem);
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   if (f->defining) return NC3_EINDEFINE;
   if (writing && !f->writable) return NC3_EPERM;
   var_t *v = &f->vars[varid];
   int esz = type_size[v->type];
   uint64_t nrec = v->is_record ? f->numrecs : 1;
   const int ranged = count != UINT64_MAX;
   if (ranged && (v->is_record || first > v->per_rec || count > v->per_rec - first)) return NC3_EEDGE;
   size_t chunk = CHUNK_ELEMS;
   unsigned char *buf = (unsigned char *) malloc (chunk * (size_t) esz);
   if (!buf) return NC3_ENOMEM;
   int range_err = 0;
   uint64_t done = 0;

   for (uint64_t r = 0; r < nrec; r++) {
      int64_t off = v->begin + (int64_t) (r * f->recsize) + (ranged ? (int64_t) (first * (uint64_t) esz) : 0);
      if (fseeko (f->fp, (off_t) off, SEEK_SET)) { free (buf); return NC3_EIO; }
      uint64_t left = ranged ? count : v->per_rec;
      while (left) {
         size_t m = left < chunk ? (size_t) left : chunk;
         if (!writing) {
            if (fread (buf, (size_t) esz, m, f->fp) != m) { free (buf); return NC3_EIO; }
            if (mt == AS_DOUBLE) {
               double *o = (double *) mem + done;
               if (v->type == NC3_DOUBLE) {      /* fast path: the big nzval / tracer arrays */
                  for (size_t e = 0; e < m; e++) {
                     uint64_t u;
                     memcpy (&u, buf + 8 * e, 8);
                     u = __builtin_bswap64 (u);
                     memcpy (&o[e], &u, 8);
                  }
               } else if (v->type == NC3_FLOAT) {   /* fast path: POP history fields are float32 */
                  for (size_t e = 0; e < m; e++) {
                     uint32_t u;
                     float x;
                     memcpy (&u, buf + 4 * e, 4);
                     u = __builtin_bswap32 (u);
                     memcpy (&x, &u, 4);
                     o[e] = (double) x;
                  }
               } else
                  for (size_t e = 0; e < m; e++) o[e] = decode_double (buf + (size_t) esz * e, v->type);
            } else {
               int *o = (int *) mem + done;
               if (v->type == NC3_INT) {
                  for (size_t e = 0; e < m; e++) {
                     uint32_t u;
                     memcpy (&u, buf + 4 * e, 4);
                     o[e] = (int) __builtin_bswap32 (u);
                  }
               } else
                  for (size_t e = 0; e < m; e++) {
                     double x = decode_double (buf + (size_t) esz * e, v->type);
                     if (!(x >= (double) INT_MIN && x <= (double) INT_MAX)) { range_err = 1; x = 0; }
                     o[e] = (int) x;
                  }
            }
         } else {
            if (mt == AS_DOUBLE) {
               const double *in = (const double *) mem + done;
               if (v->type == NC3_DOUBLE) {      /* fast path: nzval_row_wise, tracer fields */
                  for (size_t e = 0; e < m; e++) {
                     uint64_t u;
                     memcpy (&u, &in[e], 8);
                     u = __builtin_bswap64 (u);
                     memcpy (buf + 8 * e, &u, 8);
                  }
               } else
                  for (size_t e = 0; e < m; e++) range_err |= encode_double (buf + (size_t) esz * e, v->type, in[e]);
            } else {
               const int *in = (const int *) mem + done;
               if (v->type == NC3_INT) {         /* fast path: colind, rowptr, index maps */
                  for (size_t e = 0; e < m; e++) {
                     uint32_t u = __builtin_bswap32 ((uint32_t) in[e]);
                     memcpy (buf + 4 * e, &u, 4);
                  }
               } else
                  for (size_t e = 0; e < m; e++) range_err |= encode_double (buf + (size_t) esz * e, v->type, (double) in[e]);
            }
            if (fwrite (buf, (size_t) esz, m, f->fp) != m) { free (buf); return NC3_EIO; }
         }
         done += m;
         left -= m;
      }
   }
   free (buf);
   if (writing && fflush (f->fp)) return NC3_EIO;
   return range_err ? NC3_ERANGE : NC3_NOERR;
}

int nc3_get_var_double (nc3_file *f, int varid, double *out) { return transfer (f, varid, out, AS_DOUBLE, 0, 0, UINT64_MAX); }
int nc3_get_var_int (nc3_file *f, int varid, int *out) { return transfer (f, varid, out, AS_INT, 0, 0, UINT64_MAX); }
int nc3_put_var_double (nc3_file *f, int varid, const double *in) { return transfer (f, varid, (void *) in, AS_DOUBLE, 1, 0, UINT64_MAX); }
int nc3_put_var_int (nc3_file *f, int varid, const int *in) { return transfer (f, varid, (void *) in, AS_INT, 1, 0, UINT64_MAX); }
int nc3_get_vara_double (nc3_file *f, int varid, size_t first, size_t count, double *out) { return transfer (f, varid, out, AS_DOUBLE, 0, first, count); }
int nc3_get_vara_int (nc3_file *f, int varid, size_t first, size_t count, int *out) { return transfer (f, varid, out, AS_INT, 0, first, count); }

int nc3_get_att_double (nc3_file *f, int varid, const char *attname, double *val)
{
   if (f->h4) return nc4_get_att_double (f->h4, varid, attname, val);
   int n;
   att_t *a;
   if (varid == -1) { n = f->ngatts; a = f->gatts; }
   else if (varid >= 0 && varid < f->nvars) { n = f->vars[varid].natts; a = f->vars[varid].atts; }
   else return NC3_ENOTVAR;
   for (int i = 0; i < n; i++)
      if (strcmp (a[i].name, attname) == 0) {
         if (a[i].type == NC3_CHAR || a[i].n == 0) return NC3_ERANGE;
         *val = decode_double (a[i].raw, a[i].type);
         return NC3_NOERR;
      }
   return NC3_ENOTATT;
}

/* ---------------------------------------------------------------- define mode */

int nc3_create (const char *path, int version, nc3_file **out)
{
   *out = NULL;
   if (version != 1 && version != 2 && version != 5) return NC3_EINVAL;
   /* like nc_create(NC_CLOBBER): the file exists (empty) from this point on */
   FILE *fp = fopen (path, "w+b");
   if (!fp) return NC3_EIO;
   nc3_file *f = (nc3_file *) calloc (1, sizeof (nc3_file));
   if (!f) { fclose (fp); return NC3_ENOMEM; }
   f->fp = fp;
   f->path = strdup (path);
   if (!f->path) { nc3_close (f); return NC3_ENOMEM; }
   f->writable = 1;
   f->version = version;
   f->defining = 1;
   *out = f;
   return NC3_NOERR;
}

int nc3_redef (nc3_file *f)
{
   if (f->h4) return NC3_EHDF5;                 /* netCDF-4 files are read and updated in place, not extended */
   if (!f->writable) return NC3_EPERM;
   if (f->defining) return NC3_EINDEFINE;
   free (f->old_begin);
   f->old_begin = (int64_t *) malloc ((size_t) (f->nvars ? f->nvars : 1) * sizeof (int64_t));
   if (!f->old_begin) return NC3_ENOMEM;
   for (int i = 0; i < f->nvars; i++) f->old_begin[i] = f->vars[i].begin;
   f->nvars_on_disk = f->nvars;
   f->defining = 1;
   return NC3_NOERR;
}

int nc3_inq_dimid (nc3_file *f, const char *name, int *dimid)
{
   for (int i = 0; i < f->ndims; i++)
      if (strcmp (f->dims[i].name, name) == 0) { *dimid = i; return NC3_NOERR; }
   return NC3_EBADDIM;
}

int nc3_def_dim (nc3_file *f, const char *name, size_t len, int *dimid)
{
   int id;
   if (!f->defining) return NC3_ENOTINDEFINE;
   if (len == 0) return NC3_EINVAL;                 /* no new record dimension */
   if (f->version != 5 && (uint64_t) len > 0xFFFFFFFFull) return NC3_EINVAL;
   if (nc3_inq_dimid (f, name, &id) == NC3_NOERR) return NC3_ENAMEINUSE;
   dim_t *d = (dim_t *) realloc (f->dims, (size_t) (f->ndims + 1) * sizeof (dim_t));
   if (!d) return NC3_ENOMEM;
   f->dims = d;
   d[f->ndims].name = strdup (name);
   d[f->ndims].len = (uint64_t) len;
   if (!d[f->ndims].name) return NC3_ENOMEM;
   if (dimid) *dimid = f->ndims;
   f->ndims++;
   return NC3_NOERR;
}

int nc3_def_var (nc3_file *f, const char *name, int nc_type, int ndims, const int *dimids, int *varid)
{
   int id;
   if (!f->defining) return NC3_ENOTINDEFINE;
   if (nc_type < 1 || nc_type > 11 || ndims < 0 || ndims > 1024) return NC3_EINVAL;
   if (f->version != 5 && nc_type > NC3_DOUBLE) return NC3_EINVAL;
   if (nc3_inq_varid (f, name, &id) == NC3_NOERR) return NC3_ENAMEINUSE;
   for (int d = 0; d < ndims; d++) {
      if (dimids[d] < 0 || dimids[d] >= f->ndims) return NC3_EBADDIM;
      if (f->dims[dimids[d]].len == 0) return NC3_EINVAL;          /* record variable */
   }
   var_t *vs = (var_t *) realloc (f->vars, (size_t) (f->nvars + 1) * sizeof (var_t));
   if (!vs) return NC3_ENOMEM;
   f->vars = vs;
   var_t *v = &vs[f->nvars];
   memset (v, 0, sizeof (*v));
   v->name = strdup (name);
   v->dimids = (int *) calloc (ndims ? (size_t) ndims : 1, sizeof (int));
   if (!v->name || !v->dimids) { free (v->name); free (v->dimids); return NC3_ENOMEM; }
   v->ndims = ndims;
   v->type = nc_type;
   v->per_rec = 1;
   for (int d = 0; d < ndims; d++) { v->dimids[d] = dimids[d]; v->per_rec *= f->dims[dimids[d]].len; }
   if (varid) *varid = f->nvars;
   f->nvars++;
   return NC3_NOERR;
}

static int put_att_raw (nc3_file *f, int varid, const char *name, int type, uint64_t n, unsigned char *raw)
{
   int *pn;
   att_t **pa;
   if (!f->defining) { free (raw); return NC3_ENOTINDEFINE; }
   if (varid == -1) { pn = &f->ngatts; pa = &f->gatts; }
   else if (varid >= 0 && varid < f->nvars) { pn = &f->vars[varid].natts; pa = &f->vars[varid].atts; }
   else { free (raw); return NC3_ENOTVAR; }
   for (int i = 0; i < *pn; i++)
      if (strcmp ((*pa)[i].name, name) == 0) {              /* overwrite in place */
         free ((*pa)[i].raw);
         (*pa)[i].type = type;
         (*pa)[i].n = n;
         (*pa)[i].raw = raw;
         return NC3_NOERR;
      }
   att_t *a = (att_t *) realloc (*pa, (size_t) (*pn + 1) * sizeof (att_t));
   if (!a) { free (raw); return NC3_ENOMEM; }
   *pa = a;
   a[*pn].name = strdup (name);
   a[*pn].type = type;
   a[*pn].n = n;
   a[*pn].raw = raw;
   if (!a[*pn].name) { free (raw); return NC3_ENOMEM; }
   (*pn)++;
   return NC3_NOERR;
}

static unsigned char *att_buf (uint64_t nbytes)
{
   size_t padded = (size_t) ((nbytes + 3) & ~(uint64_t) 3);
   return (unsigned char *) calloc (padded ? padded : 1, 1);
}

int nc3_put_att_text (nc3_file *f, int varid, const char *name, size_t len, const char *text)
{
   unsigned char *raw = att_buf (len);
   if (!raw) return NC3_ENOMEM;
   memcpy (raw, text, len);
   return put_att_raw (f, varid, name, NC3_CHAR, len, raw);
}

int nc3_put_att_double (nc3_file *f, int varid, const char *name, int nc_type, size_t n, const double *vals)
{
   if (nc_type < 1 || nc_type > 11 || nc_type == NC3_CHAR) return NC3_EINVAL;
   int esz = type_size[nc_type];
   unsigned char *raw = att_buf ((uint64_t) n * (uint64_t) esz);
   if (!raw) return NC3_ENOMEM;
   int bad = 0;
   for (size_t e = 0; e < n; e++) bad |= encode_double (raw + (size_t) esz * e, nc_type, vals[e]);
   int status = put_att_raw (f, varid, name, nc_type, n, raw);
   return status ? status : (bad ? NC3_ERANGE : NC3_NOERR);
}

int nc3_put_att_int (nc3_file *f, int varid, const char *name, int nc_type, size_t n, const int *vals)
{
   double *tmp = (double *) malloc ((n ? n : 1) * sizeof (double));
   if (!tmp) return NC3_ENOMEM;
   for (size_t e = 0; e < n; e++) tmp[e] = (double) vals[e];
   int status = nc3_put_att_double (f, varid, name, nc_type, n, tmp);
   free (tmp);
   return status;
}

/* ---- header writer */

typedef struct { FILE *fp; int ok; int wide; uint64_t bytes; } wr_t;

static void wr_bytes (wr_t *w, const void *p, size_t n)
{
   if (w->fp && n && fwrite (p, 1, n, w->fp) != n) w->ok = 0;
   w->bytes += n;
}

static void wr_u32 (wr_t *w, uint32_t u)
{
   unsigned char b[4];
   store_be (b, 4, u);
   wr_bytes (w, b, 4);
}

static void wr_u64 (wr_t *w, uint64_t u)
{
   unsigned char b[8];
   store_be (b, 8, u);
   wr_bytes (w, b, 8);
}

static void wr_nonneg (wr_t *w, uint64_t u) { if (w->wide) wr_u64 (w, u); else wr_u32 (w, (uint32_t) u); }

static void wr_name (wr_t *w, const char *s)
{
   static const unsigned char zero[4] = { 0, 0, 0, 0 };
   size_t n = strlen (s);
   wr_nonneg (w, n);
   wr_bytes (w, s, n);
   wr_bytes (w, zero, (4 - (n & 3)) & 3);
}

static void wr_attlist (wr_t *w, int natts, const att_t *a)
{
   if (natts == 0) { wr_u32 (w, 0); wr_nonneg (w, 0); return; }
   wr_u32 (w, TAG_ATT);
   wr_nonneg (w, (uint64_t) natts);
   for (int i = 0; i < natts; i++) {
      wr_name (w, a[i].name);
      wr_u32 (w, (uint32_t) a[i].type);
      wr_nonneg (w, a[i].n);
      uint64_t nbytes = a[i].n * (uint64_t) type_size[a[i].type];
      wr_bytes (w, a[i].raw, (size_t) ((nbytes + 3) & ~(uint64_t) 3));
   }
}

static uint64_t var_bytes (const var_t *v)
{
   uint64_t raw = v->per_rec * (uint64_t) type_size[v->type];
   return (raw + 3) & ~(uint64_t) 3;
}

/* fp == NULL: size the header only */
static uint64_t write_header (nc3_file *f, FILE *fp, int *ok)
{
   wr_t w = { fp, 1, f->version == 5, 0 };
   unsigned char magic[4] = { 'C', 'D', 'F', (unsigned char) f->version };
   wr_bytes (&w, magic, 4);
   wr_nonneg (&w, f->numrecs);
   if (f->ndims == 0) { wr_u32 (&w, 0); wr_nonneg (&w, 0); }
   else {
      wr_u32 (&w, TAG_DIM);
      wr_nonneg (&w, (uint64_t) f->ndims);
      for (int i = 0; i < f->ndims; i++) { wr_name (&w, f->dims[i].name); wr_nonneg (&w, f->dims[i].len); }
   }
   wr_attlist (&w, f->ngatts, f->gatts);
   if (f->nvars == 0) { wr_u32 (&w, 0); wr_nonneg (&w, 0); }
   else {
      wr_u32 (&w, TAG_VAR);
      wr_nonneg (&w, (uint64_t) f->nvars);
      for (int i = 0; i < f->nvars; i++) {
         const var_t *v = &f->vars[i];
         wr_name (&w, v->name);
         wr_nonneg (&w, (uint64_t) v->ndims);
         for (int d = 0; d < v->ndims; d++) wr_nonneg (&w, (uint64_t) v->dimids[d]);
         wr_attlist (&w, v->natts, v->atts);
         wr_u32 (&w, (uint32_t) v->type);
         uint64_t vs = var_bytes (v);
         if (!w.wide && vs > 0xFFFFFFFFull) vs = 0xFFFFFFFFull;
         wr_nonneg (&w, vs);
         if (f->version == 1) wr_u32 (&w, (uint32_t) v->begin); else wr_u64 (&w, (uint64_t) v->begin);
      }
   }
   if (ok) *ok = w.ok;
   return w.bytes;
}

static void default_fill (int type, unsigned char *p)
{
   switch (type) {
   case NC3_BYTE: p[0] = (unsigned char) (int8_t) -127; break;
   case NC3_CHAR: p[0] = 0; break;
   case NC3_SHORT: store_be (p, 2, (uint64_t) (uint16_t) (int16_t) -32767); break;
   case NC3_INT: store_be (p, 4, (uint64_t) (uint32_t) (int32_t) -2147483647); break;
   case NC3_FLOAT: { float x = 9.9692099683868690e+36f; uint32_t u; memcpy (&u, &x, 4); store_be (p, 4, u); break; }
   case NC3_DOUBLE: { double x = 9.9692099683868690e+36; uint64_t u; memcpy (&u, &x, 8); store_be (p, 8, u); break; }
   case NC3_UBYTE: p[0] = 255; break;
   case NC3_USHORT: store_be (p, 2, 65535u); break;
   case NC3_UINT: store_be (p, 4, 4294967295u); break;
   case NC3_INT64: store_be (p, 8, (uint64_t) (int64_t) -9223372036854775806LL); break;
   case NC3_UINT64: store_be (p, 8, 18446744073709551614ULL); break;
   }
}

int nc3_set_fill (nc3_file *f, int fill)
{
   f->nofill = !fill;
   return NC3_NOERR;
}

/* NC_NOFILL: reserve the variable's bytes without writing them (a hole the caller fills right away) */
static int skip_var (FILE *fp, const var_t *v)
{
   uint64_t nbytes = var_bytes (v);
   static const unsigned char zero[1] = { 0 };
   if (nbytes == 0) return NC3_NOERR;
   if (fseeko (fp, (off_t) (nbytes - 1), SEEK_CUR)) return NC3_EIO;
   return fwrite (zero, 1, 1, fp) == 1 ? NC3_NOERR : NC3_EIO;
}

static int fill_var (FILE *fp, const var_t *v)
{
   int esz = type_size[v->type];
   unsigned char fv[8];
   default_fill (v->type, fv);
   for (int i = 0; i < v->natts; i++)
      if (strcmp (v->atts[i].name, "_FillValue") == 0 && v->atts[i].type == v->type && v->atts[i].n >= 1)
         memcpy (fv, v->atts[i].raw, (size_t) esz);
   size_t chunk = CHUNK_ELEMS;
   uint64_t total = v->per_rec;
   if (total < chunk) chunk = (size_t) (total ? total : 1);
   unsigned char *buf = (unsigned char *) malloc (chunk * (size_t) esz);
   if (!buf) return NC3_ENOMEM;
   for (size_t e = 0; e < chunk; e++) memcpy (buf + e * (size_t) esz, fv, (size_t) esz);
   uint64_t left = total;
   int status = NC3_NOERR;
   while (left && !status) {
      size_t m = left < chunk ? (size_t) left : chunk;
      if (fwrite (buf, (size_t) esz, m, fp) != m) status = NC3_EIO;
      left -= m;
   }
   free (buf);
   uint64_t pad = var_bytes (v) - total * (uint64_t) esz;
   static const unsigned char zero[4] = { 0, 0, 0, 0 };
   if (!status && pad && fwrite (zero, 1, (size_t) pad, fp) != pad) status = NC3_EIO;
   return status;
}

static int copy_bytes (FILE *src, int64_t src_off, FILE *dst, uint64_t nbytes)
{
   size_t chunk = (size_t) 8 << 20;
   if (nbytes < chunk) chunk = (size_t) (nbytes ? nbytes : 1);
   unsigned char *buf = (unsigned char *) malloc (chunk);
   if (!buf) return NC3_ENOMEM;
   int status = NC3_NOERR;
   if (fseeko (src, (off_t) src_off, SEEK_SET)) status = NC3_EIO;
   while (nbytes && !status) {
      size_t m = nbytes < chunk ? (size_t) nbytes : chunk;
      if (fread (buf, 1, m, src) != m || fwrite (buf, 1, m, dst) != m) status = NC3_EIO;
      nbytes -= m;
   }
   free (buf);
   return status;
}

int nc3_enddef (nc3_file *f)
{
   if (!f->defining) return NC3_ENOTINDEFINE;

   /* layout: header, fixed-size variables in definition order, then the record section */
   uint64_t off = (write_header (f, NULL, NULL) + 3) & ~(uint64_t) 3;
   int64_t old_rec_begin = -1;
   for (int i = 0; i < f->nvars; i++)
      if (!f->vars[i].is_record) { f->vars[i].begin = (int64_t) off; off += var_bytes (&f->vars[i]); }
   uint64_t rec_begin = off;
   for (int i = 0; i < f->nvars; i++)
      if (f->vars[i].is_record) {
         if (old_rec_begin < 0 || f->old_begin[i] < old_rec_begin) old_rec_begin = f->old_begin[i];
         f->vars[i].begin = (int64_t) off;
         off += (f->recsize == f->vars[i].per_rec * (uint64_t) type_size[f->vars[i].type]) ? f->recsize : var_bytes (&f->vars[i]);
      }
   if (f->version == 1)
      for (int i = 0; i < f->nvars; i++)
         if ((uint64_t) f->vars[i].begin > 0x7FFFFFFFull) return NC3_EINVAL;

   /* write the new image beside the old one, then swap */
   size_t plen = strlen (f->path);
   char *tmp = (char *) malloc (plen + 16);
   if (!tmp) return NC3_ENOMEM;
   snprintf (tmp, plen + 16, "%s.nc3tmp", f->path);
   FILE *out = fopen (tmp, "w+b");
   if (!out) { free (tmp); return NC3_EIO; }
   int ok = 1;
   int status = NC3_NOERR;
   uint64_t hbytes = write_header (f, out, &ok);
   if (!ok) status = NC3_EIO;
   static const unsigned char zero[4] = { 0, 0, 0, 0 };
   if (!status && (hbytes & 3) && fwrite (zero, 1, 4 - (hbytes & 3), out) != 4 - (hbytes & 3)) status = NC3_EIO;
   for (int i = 0; i < f->nvars && !status; i++) {
      const var_t *v = &f->vars[i];
      if (v->is_record) continue;
      if (i < f->nvars_on_disk && f->old_begin) status = copy_bytes (f->fp, f->old_begin[i], out, var_bytes (v));
      else status = f->nofill ? skip_var (out, v) : fill_var (out, v);
   }
   if (!status && old_rec_begin >= 0 && f->numrecs) {
      (void) rec_begin;
      status = copy_bytes (f->fp, old_rec_begin, out, f->numrecs * f->recsize);
   }
   if (!status && fflush (out)) status = NC3_EIO;
   if (status) { fclose (out); remove (tmp); free (tmp); return status; }
   fclose (f->fp);
   f->fp = out;
   if (rename (tmp, f->path)) status = NC3_EIO;
   free (tmp);
   free (f->old_begin);
   f->old_begin = NULL;
   f->nvars_on_disk = f->nvars;
   f->defining = 0;
   return status;
}
