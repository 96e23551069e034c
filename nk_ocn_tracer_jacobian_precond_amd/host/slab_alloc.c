/* 2-D / 3-D array allocators: a pointer ladder over ONE contiguous slab so that field[0]
 * (2-D) or field[0][0] (3-D) can be handed to whole-variable I/O.  Same contract as
 * reference src/memory.c:11-155 (NULL on failure; free_* takes the ladder's top pointer). */
#include <stdlib.h>

#include "nkp_host.h"

/* ladder for a [n0][n1] view over `slab` of elements of `esz` bytes */
static void **rows_over (void *slab, size_t n0, size_t n1, size_t esz)
{
   void **rows = (void **) malloc ((n0 ? n0 : 1) * sizeof (void *));
   if (!rows) return NULL;
   for (size_t r = 0; r < n0; r++) rows[r] = (char *) slab + r * n1 * esz;
   return rows;
}

static void **alloc_2d (int n0, int n1, size_t esz)
{
   if (n0 <= 0 || n1 <= 0) return NULL;
   void *slab = malloc ((size_t) n0 * (size_t) n1 * esz);
   if (!slab) return NULL;
   void **rows = rows_over (slab, (size_t) n0, (size_t) n1, esz);
   if (!rows) free (slab);
   return rows;
}

static void ***alloc_3d (int n0, int n1, int n2, size_t esz)
{
   if (n0 <= 0 || n1 <= 0 || n2 <= 0) return NULL;
   void *slab = malloc ((size_t) n0 * (size_t) n1 * (size_t) n2 * esz);
   if (!slab) return NULL;
   void **rows = rows_over (slab, (size_t) n0 * (size_t) n1, (size_t) n2, esz);   /* [n0*n1] row pointers */
   if (!rows) { free (slab); return NULL; }
   void ***planes = (void ***) malloc ((size_t) n0 * sizeof (void **));
   if (!planes) { free (rows); free (slab); return NULL; }
   for (int p = 0; p < n0; p++) planes[p] = rows + (size_t) p * (size_t) n1;
   return planes;
}

int **malloc_2d_int (int jmt_, int imt_) { return (int **) alloc_2d (jmt_, imt_, sizeof (int)); }
double **malloc_2d_double (int jmt_, int imt_) { return (double **) alloc_2d (jmt_, imt_, sizeof (double)); }
int ***malloc_3d_int (int km_, int jmt_, int imt_) { return (int ***) alloc_3d (km_, jmt_, imt_, sizeof (int)); }
double ***malloc_3d_double (int km_, int jmt_, int imt_) { return (double ***) alloc_3d (km_, jmt_, imt_, sizeof (double)); }

static void release_2d (void **rows) { if (rows) { free (rows[0]); free (rows); } }
static void release_3d (void ***planes) { if (planes) { free (planes[0][0]); free (planes[0]); free (planes); } }

void free_2d_int (int **ptr) { release_2d ((void **) ptr); }
void free_2d_double (double **ptr) { release_2d ((void **) ptr); }
void free_3d_int (int ***ptr) { release_3d ((void ***) ptr); }
void free_3d_double (double ***ptr) { release_3d ((void ***) ptr); }
