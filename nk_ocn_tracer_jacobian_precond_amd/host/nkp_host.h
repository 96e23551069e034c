/* Host-side mirror of the reference's C surface on the solve path.
 *
 * Same names, argument meaning, return convention (0 = ok, non-zero = failure with a message
 * on stderr prefixed "(iam)") and the same global-state model as the reference, so that the
 * solve_ABglobal / solve_ABdist mains and the parity tests read like the reference's own:
 *
 *   file_io   : reference src/file_io.h:6-26   (implemented on nc3_codec instead of libnetcdf)
 *   memory    : reference src/memory.h:6-14    (one contiguous slab behind a pointer ladder)
 *   misc      : reference src/misc.h:6-8
 *   globals   : reference src/globals.h:6-7
 *   grid      : reference src/grid.h:6-31      (dims for the solvers; loader + writer for gen_A)
 *   matrix    : reference src/matrix.h:6-81    (readers + index maps for the solvers; generator,
 *               option types/globals and writers for gen_A)
 *
 * Pure C, no GPU dependency: this library loads on any host (CPU tests use it directly).
 */
#ifndef NKP_HOST_H
#define NKP_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the reference borrows int_t from SuperLU_DIST (README:14-15); the file schema is int32 */
typedef int int_t;

/* ---- globals (reference src/globals.h:6-7) ------------------------------------------- */
extern int dbg_lvl;
extern int iam;

/* ---- misc (reference src/misc.c:11-95) ------------------------------------------------ */
int parse_to_long (char *str, long *val);
int parse_to_int (char *str, int *val);
int parse_to_double (char *str, double *val);

/* ---- memory (reference src/memory.c:11-155) ------------------------------------------- */
int **malloc_2d_int (int jmt, int imt);
void free_2d_int (int **ptr);
int ***malloc_3d_int (int km, int jmt, int imt);
void free_3d_int (int ***ptr);
double **malloc_2d_double (int jmt, int imt);
void free_2d_double (double **ptr);
double ***malloc_3d_double (int km, int jmt, int imt);
void free_3d_double (double ***ptr);

/* ---- file_io (reference src/file_io.c:10-368) ----------------------------------------- */
int handle_nc_error (char *subname, char *cdf_subname, char *msg, int status);
int var_exists_in_file (char *fname, char *varname, int *retval);
int get_att_double (char *fname, char *varname, char *attname, double *val);
/* additions of this build: a range of a 1-D variable */
int get_vara_1d_int (char *fname, char *varname, size_t first, size_t count, int *field);
int get_vara_1d_double (char *fname, char *varname, size_t first, size_t count, double *field);
int get_var_1d_int (char *fname, char *varname, int *field);
int get_var_2d_int (char *fname, char *varname, int **field);
int get_var_3d_int (char *fname, char *varname, int ***field);
int put_var_1d_int (char *fname, char *varname, int *field);
int put_var_2d_int (char *fname, char *varname, int **field);
int put_var_3d_int (char *fname, char *varname, int ***field);
int get_var_1d_double (char *fname, char *varname, double *field);
int get_var_2d_double (char *fname, char *varname, double **field);
int get_var_3d_double (char *fname, char *varname, double ***field);
int put_var_1d_double (char *fname, char *varname, double *field);
int put_var_2d_double (char *fname, char *varname, double **field);
int put_var_3d_double (char *fname, char *varname, double ***field);

/* ---- grid dims (reference src/grid.c:33-86) -------------------------------------------- */
extern int imt;
extern int jmt;
extern int km;
int get_grid_dims (char *fname);

/* ---- grid info for the matrix generator (reference src/grid.c:90-330, src/grid.h:15-31) -- */
extern char *circ_fname;
extern char *reg_fname;
extern double *z_t;
extern double *dz;
extern double **TLONG;
extern double **TLAT;
extern int **KMT;
extern int **KMU;
extern double **TAREA;
int get_grid_info (char *circ_fname, char *reg_fname);
int put_grid_info (char *fname);
void free_grid_info (void);

/* ---- matrix + index maps (reference src/matrix.c:373-464, 3943-4070) ------------------- */
typedef struct { int i; int j; int k; } int3;

extern int tracer_state_len;
extern int ***int3_to_tracer_state_ind;
extern int3 *tracer_state_ind_to_int3;
extern int coupled_tracer_cnt;
extern int flat_len;
extern int nnz;
extern double *nzval_row_wise;
extern int_t *colind;
extern int_t *rowptr;

int get_ind_maps (char *fname);
void free_ind_maps (void);
int get_sparse_matrix (char *fname);
/* additions of this build (row-distributed flavour: every rank reads its own rows, not the whole matrix):
 * get_sparse_matrix_header = dims, coupled_tracer_cnt and the row pointers only (colind / nzval_row_wise stay NULL);
 * get_sparse_matrix_rows = the entries of rows [row0, row1) into caller arrays of rowptr[row1] - rowptr[row0] elements,
 * with the same structural checks get_sparse_matrix applies. */
int get_sparse_matrix_header (char *fname);
int get_sparse_matrix_rows (char *fname, int row0, int row1, int_t *colind_out, double *nzval_out);
void free_sparse_matrix (void);

/* ---- matrix generator (reference src/matrix.h:6-9,26-56,70-81; src/matrix.c:163-369, 466-3939) */
typedef enum { adv_none, adv_donor, adv_cent, adv_upwind3 } adv_opt_t;
typedef enum { hmix_none, hmix_const, hmix_hor_file, hmix_isop_file } hmix_opt_t;
typedef enum { vmix_none, vmix_const, vmix_file, vmix_matrix_file } vmix_opt_t;
typedef enum { sink_none, sink_const, sink_const_shallow, sink_file, sink_generic_tracer } sink_opt_t;
typedef struct {
   sink_opt_t sink_opt;
   double sink_rate;            /* loss rate, 1/yr */
   double sink_depth;           /* depth threshold of sink_const_shallow, cm (as z_t) */
   char *sink_field_name;
   char *sink_generic_tracer_name;
   int sink_generic_tracer_depends_layer_cnt;
   char *pv_field_name;
   char *d_SF_d_TRACER_field_name;
} per_tracer_opt_t;
typedef enum { coupled_tracer_none, coupled_tracer_OCMIP_BGC_PO4_DOP, coupled_tracer_DIC_SHADOW_ALK_SHADOW } coupled_tracer_opt_t;

extern adv_opt_t adv_opt;
extern int l_adv_enforce_divfree;
extern hmix_opt_t hmix_opt;
extern vmix_opt_t vmix_opt;
extern char *tracer_fname;
extern per_tracer_opt_t *per_tracer_opt;
extern coupled_tracer_opt_t coupled_tracer_opt;

int gen_ind_maps (void);
int put_ind_maps (char *fname);
int gen_sparse_matrix (double day_cnt);
int put_sparse_matrix (char *fname);

/* ---- additions of this build (no reference counterpart) ------------------------------- */

/* Water-column boundaries of the flat state vector, derived from the index maps: a new
 * column starts wherever tracer_state_ind_to_int3[s].k == 0 (ordering: reference
 * src/matrix.c:239-251).  With coupled tracers (tracer-major rows, src/matrix.c:778-784)
 * the pattern repeats per tracer.  Returns a malloc'd array of *nblk + 1 row offsets. */
int_t *nkp_column_blocks (int *nblk);
/* Grid position (i, j) of each of those blocks (its first row's index-map entry); 0 = ok. */
int nkp_column_coords (int nblk, int *col_i, int *col_j);

/* B[t*tracer_state_len + s] = field[k_s][j_s][i_s]  (reference src/solve_ABglobal.c:184-191)
 * and its inverse which leaves every non-ocean value of `field` untouched (:242-248). */
void nkp_flatten_tracer (int tracer_ind, double ***field_3d, double *B);
void nkp_unflatten_tracer (int tracer_ind, const double *B, double ***field_3d);

/* Total element count of a variable (records included); 0 = ok. */
int nkp_var_nelems (char *fname, char *varname, size_t *nelems);

/* Contiguous row-block partition of the reference's distributed solver
 * (src/solve_ABdist.c:141-144): m_loc = n / nprocs, the last rank takes the remainder. */
void nkp_rowblock_partition (int n, int nprocs, int rank, int *fst_row, int *m_loc);
/* ... with every cut snapped to the nearest water-column boundary (what the GPU solver needs). */
void nkp_rowblock_partition_snapped (const int_t *blk_start, int nblk, int nprocs, int rank, int *fst_row, int *m_loc, int *fst_blk, int *nblk_loc);

#ifdef __cplusplus
}
#endif
#endif
