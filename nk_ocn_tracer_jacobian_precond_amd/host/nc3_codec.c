/* NetCDF classic-format codec: see nc3_codec.h.  Written from the classic-format grammar
 * (header = magic numrecs dim_list gatt_list var_list; big-endian; names and attribute
 * payloads padded to 4 bytes; record variables interleaved per record). */
#define _FILE_OFFSET_BITS 64
#include "nc3_codec.h"

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
   FILE *fp;
   int writable;
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
   if (f->fp && fclose (f->fp)) status = NC3_EIO;
   for (int i = 0; i < f->ndims; i++) free (f->dims[i].name);
   free (f->dims);
   for (int i = 0; i < f->nvars; i++) {
      free (f->vars[i].name);
      free (f->vars[i].dimids);
      free_atts (f->vars[i].natts, f->vars[i].atts);
   }
   free (f->vars);
   free_atts (f->ngatts, f->gatts);
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
   if (magic[0] == 0x89 && magic[1] == 'H' && magic[2] == 'D' && magic[3] == 'F') { fclose (fp); return NC3_EHDF5; }
   if (magic[0] != 'C' || magic[1] != 'D' || magic[2] != 'F' || (magic[3] != 1 && magic[3] != 2 && magic[3] != 5)) {
      fclose (fp);
      return NC3_ENOTNC;
   }
   nc3_file *f = (nc3_file *) calloc (1, sizeof (nc3_file));
   if (!f) { fclose (fp); return NC3_ENOMEM; }
   f->fp = fp;
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
   if (!r.ok) { status = NC3_ENOTNC; goto fail; }

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
   case NC3_EHDF5: return "NetCDF: netCDF-4/HDF5 files are not supported by this build (classic CDF-1/2/5 only)";
   case NC3_ENOTATT: return "NetCDF: Attribute not found";
   case NC3_EIO: return "NetCDF: I/O failure (open, seek, read or write)";
   case NC3_EPERM: return "NetCDF: Write to read only";
   default: return "NetCDF: Unknown error";
   }
}

/* ---------------------------------------------------------------- inquiries */

int nc3_inq_dimlen (nc3_file *f, const char *dimname, size_t *len)
{
   for (int i = 0; i < f->ndims; i++)
      if (strcmp (f->dims[i].name, dimname) == 0) {
         *len = (size_t) (f->dims[i].len ? f->dims[i].len : f->numrecs);
         return NC3_NOERR;
      }
   return NC3_EBADDIM;
}

int nc3_inq_varid (nc3_file *f, const char *varname, int *varid)
{
   for (int i = 0; i < f->nvars; i++)
      if (strcmp (f->vars[i].name, varname) == 0) { *varid = i; return NC3_NOERR; }
   return NC3_ENOTVAR;
}

int nc3_inq_var (nc3_file *f, int varid, int *nc_type, int *ndims, size_t *nelems)
{
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   var_t *v = &f->vars[varid];
   if (nc_type) *nc_type = v->type;
   if (ndims) *ndims = v->ndims;
   if (nelems) *nelems = (size_t) (v->per_rec * (v->is_record ? f->numrecs : 1));
   return NC3_NOERR;
}

int nc3_inq_var_dimlens (nc3_file *f, int varid, size_t *dimlens)
{
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

static int transfer (nc3_file *f, int varid, void *mem, mem_t mt, int writing)
{
   if (varid < 0 || varid >= f->nvars) return NC3_ENOTVAR;
   if (writing && !f->writable) return NC3_EPERM;
   var_t *v = &f->vars[varid];
   int esz = type_size[v->type];
   uint64_t nrec = v->is_record ? f->numrecs : 1;
   size_t chunk = CHUNK_ELEMS;
   unsigned char *buf = (unsigned char *) malloc (chunk * (size_t) esz);
   if (!buf) return NC3_ENOMEM;
   int range_err = 0;
   uint64_t done = 0;

   for (uint64_t r = 0; r < nrec; r++) {
      int64_t off = v->begin + (int64_t) (r * f->recsize);
      if (fseeko (f->fp, (off_t) off, SEEK_SET)) { free (buf); return NC3_EIO; }
      uint64_t left = v->per_rec;
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
               for (size_t e = 0; e < m; e++) range_err |= encode_double (buf + (size_t) esz * e, v->type, in[e]);
            } else {
               const int *in = (const int *) mem + done;
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

int nc3_get_var_double (nc3_file *f, int varid, double *out) { return transfer (f, varid, out, AS_DOUBLE, 0); }
int nc3_get_var_int (nc3_file *f, int varid, int *out) { return transfer (f, varid, out, AS_INT, 0); }
int nc3_put_var_double (nc3_file *f, int varid, const double *in) { return transfer (f, varid, (void *) in, AS_DOUBLE, 1); }
int nc3_put_var_int (nc3_file *f, int varid, const int *in) { return transfer (f, varid, (void *) in, AS_INT, 1); }

int nc3_get_att_double (nc3_file *f, int varid, const char *attname, double *val)
{
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
