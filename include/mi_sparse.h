/*
 * mi_sparse.h -- C ABI of libmi_sparse.so, the MI355X (gfx950) sparse-matmul backend.
 *
 * This is the drop-in boundary for the dot_product_mkl / gram_matrix_mkl hot path of
 * flatironinstitute/sparse_dot (sparse_dot_mkl 0.9.6).  The reference is a ctypes wrapper over
 * Intel MKL's inspector-executor Sparse BLAS + CBLAS; every entry point below replaces ONE MKL
 * symbol the reference binds in sparse_dot_mkl/_mkl_interface/_cfunctions.py (cited per
 * function) and keeps that symbol's argument order, operation / layout codes and status codes,
 * so that a maintainer can re-point the reference's `class MKL` symbol table at this library
 * (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions
 *   - Plain C: opaque handle, pointers, sizes.  No C++ or torch types cross the boundary.
 *   - Every routine returns mi_sparse_status_t (same numeric values as MKL's sparse_status_t,
 *     reference _constants.py:2-10).  Nothing throws or aborts across the boundary;
 *     mi_sparse_last_error() gives a human-readable reason for the calling thread's last failure.
 *   - POINTER LOCATION IS AUTO-DETECTED.  Any array argument (CSR arrays, dense B / C) may be a
 *     host pointer (numpy memory -- what the reference passes) or a device pointer (HBM-resident,
 *     e.g. torch tensor .data_ptr()).  Host arrays are staged through device buffers owned by the
 *     library; device arrays are used in place (zero copy).
 *   - Index width: functions that take or return index ARRAYS come in two flavours, `name`
 *     (32-bit indices, MKL LP64) and `name_64` (64-bit indices, MKL ILP64; same suffix MKL's own
 *     ILP64 API uses).  Scalar sizes (rows, cols, ld*) are always int64_t.
 *   - Value types: s = float, d = double, c = complex float, z = complex double.
 *   - All work is enqueued on the stream set by mi_sparse_set_stream() (default: the null
 *     stream).  Routines whose outputs are host pointers synchronise that stream before
 *     returning; routines whose outputs are device pointers return asynchronously.
 *   - Thread safety: calls on DIFFERENT handles may run concurrently from several host threads
 *     (ctypes releases the GIL); device / stream selection is per host thread.
 *   - HIP is initialised lazily on the first call that needs a device (never at dlopen), which
 *     keeps `import` fork-safe like the reference's KMP_INIT_AT_FORK care
 *     (reference _mkl_interface/__init__.py:3-8).
 */
#ifndef MI_SPARSE_H
#define MI_SPARSE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (== MKL sparse_status_t; reference _constants.py:2-10) ------------------- */
typedef int mi_sparse_status_t;
#define MI_SPARSE_STATUS_SUCCESS 0
#define MI_SPARSE_STATUS_NOT_INITIALIZED 1 /* NULL / destroyed handle, NULL array            */
#define MI_SPARSE_STATUS_ALLOC_FAILED 2    /* hipMalloc / host allocation failure             */
#define MI_SPARSE_STATUS_INVALID_VALUE 3   /* bad dimension, code, or dimension mismatch      */
#define MI_SPARSE_STATUS_EXECUTION_FAILED 4 /* HIP runtime error during a launch / copy       */
#define MI_SPARSE_STATUS_INTERNAL_ERROR 5
#define MI_SPARSE_STATUS_NOT_SUPPORTED 6   /* valid MKL usage this build does not implement   */

/* ---- operation / layout / misc codes (== MKL; reference _constants.py:13-57) ---------------- */
#define MI_SPARSE_OPERATION_NON_TRANSPOSE 10
#define MI_SPARSE_OPERATION_TRANSPOSE 11
#define MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE 12
#define MI_SPARSE_LAYOUT_ROW_MAJOR 101
#define MI_SPARSE_LAYOUT_COLUMN_MAJOR 102
#define MI_CBLAS_NO_TRANS 111
#define MI_CBLAS_TRANS 112
#define MI_CBLAS_CONJ_TRANS 113
#define MI_CBLAS_UPPER 121
#define MI_CBLAS_LOWER 122
#define MI_SPARSE_INDEX_BASE_ZERO 0
#define MI_SPARSE_INDEX_BASE_ONE 1
#define MI_SPARSE_MATRIX_TYPE_GENERAL 20
#define MI_SPARSE_MATRIX_TYPE_SYMMETRIC 21
#define MI_SPARSE_FILL_MODE_LOWER 40
#define MI_SPARSE_FILL_MODE_UPPER 41
#define MI_SPARSE_DIAG_NON_UNIT 50
/* stages of the two-stage product (== MKL sparse_request_t; reference _constants.py:49-53) */
#define MI_SPARSE_STAGE_FULL_MULT 90
#define MI_SPARSE_STAGE_NNZ_COUNT 91
#define MI_SPARSE_STAGE_FINALIZE_MULT 92
#define MI_SPARSE_STAGE_FULL_MULT_NO_VAL 93
#define MI_SPARSE_STAGE_FINALIZE_MULT_NO_VAL 94

/* opaque handle (== MKL sparse_matrix_t; reference _structs.py:5-9) */
struct mi_sparse_matrix;
typedef struct mi_sparse_matrix *mi_sparse_matrix_t;

/* == MKL struct matrix_descr, passed BY VALUE (reference _structs.py:13-30).  Only
 * {type = GENERAL(20), mode = 0, diag = 0} -- what the reference always passes on the hot path -- is accepted
 * there; mi_sparse_sypr takes {SYMMETRIC(21), FILL_MODE_UPPER(41), DIAG_NON_UNIT(50)} for B
 * (reference _sparse_sypr.py:104-108). */
struct mi_matrix_descr {
    int type;
    int mode;
    int diag;
};

/* == MKL_Complex8 / MKL_Complex16 (reference _structs.py:36-58), passed by value */
typedef struct { float real, imag; } mi_complex8;
typedef struct { double real, imag; } mi_complex16;

/* ============================================================================================
 * Handle life cycle
 * ========================================================================================== */

/* mkl_sparse_?_create_csr (reference _cfunctions.py:526-536; call site _common.py:310-319).
 * 4-array CSR: row i owns entries [rows_start[i], rows_end[i]).  The reference always passes
 * indptr[:-1] / indptr[1:]; when rows_end == rows_start + 1 the arrays are treated as one
 * (rows+1)-long indptr (zero copy for device pointers), otherwise they are compacted.
 * `base` 0 or 1.  Host arrays are COPIED to the device at creation (the inspector stage: H2D +
 * nnz-balanced partition metadata); device arrays are aliased and must outlive the handle.
 * Unsorted and duplicate column indices are allowed (duplicates are summed by every product). */
mi_sparse_status_t mi_sparse_s_create_csr(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *rows_start, const int32_t *rows_end,
                                          const int32_t *col_indx, const float *values);
mi_sparse_status_t mi_sparse_d_create_csr(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *rows_start, const int32_t *rows_end,
                                          const int32_t *col_indx, const double *values);
mi_sparse_status_t mi_sparse_c_create_csr(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *rows_start, const int32_t *rows_end,
                                          const int32_t *col_indx, const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_csr(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *rows_start, const int32_t *rows_end,
                                          const int32_t *col_indx, const mi_complex16 *values);
mi_sparse_status_t mi_sparse_s_create_csr_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *rows_start, const int64_t *rows_end,
                                             const int64_t *col_indx, const float *values);
mi_sparse_status_t mi_sparse_d_create_csr_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *rows_start, const int64_t *rows_end,
                                             const int64_t *col_indx, const double *values);
mi_sparse_status_t mi_sparse_c_create_csr_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *rows_start, const int64_t *rows_end,
                                             const int64_t *col_indx, const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_csr_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *rows_start, const int64_t *rows_end,
                                             const int64_t *col_indx, const mi_complex16 *values);

/* mkl_sparse_?_create_csc (reference _cfunctions.py:526-536, _common.py:272-281): same argument
 * meaning with columns in place of rows (cols_start / cols_end / row_indx). */
mi_sparse_status_t mi_sparse_s_create_csc(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *cols_start, const int32_t *cols_end,
                                          const int32_t *row_indx, const float *values);
mi_sparse_status_t mi_sparse_d_create_csc(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *cols_start, const int32_t *cols_end,
                                          const int32_t *row_indx, const double *values);
mi_sparse_status_t mi_sparse_c_create_csc(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *cols_start, const int32_t *cols_end,
                                          const int32_t *row_indx, const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_csc(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                          const int32_t *cols_start, const int32_t *cols_end,
                                          const int32_t *row_indx, const mi_complex16 *values);
mi_sparse_status_t mi_sparse_s_create_csc_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *cols_start, const int64_t *cols_end,
                                             const int64_t *row_indx, const float *values);
mi_sparse_status_t mi_sparse_d_create_csc_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *cols_start, const int64_t *cols_end,
                                             const int64_t *row_indx, const double *values);
mi_sparse_status_t mi_sparse_c_create_csc_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *cols_start, const int64_t *cols_end,
                                             const int64_t *row_indx, const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_csc_64(mi_sparse_matrix_t *A, int base, int64_t rows, int64_t cols,
                                             const int64_t *cols_start, const int64_t *cols_end,
                                             const int64_t *row_indx, const mi_complex16 *values);

/* mkl_sparse_?_create_bsr (reference _cfunctions.py:539-551; call site _common.py:363-378).
 * rows / cols count BLOCKS; blocks are block_size x block_size, stored row-major
 * (block_layout 101) or column-major (102).  The handle is expanded to element CSR on the device
 * (every stored block element becomes a structural entry). */
mi_sparse_status_t mi_sparse_s_create_bsr(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                          int64_t cols, int64_t block_size, const int32_t *rows_start,
                                          const int32_t *rows_end, const int32_t *col_indx, const float *values);
mi_sparse_status_t mi_sparse_d_create_bsr(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                          int64_t cols, int64_t block_size, const int32_t *rows_start,
                                          const int32_t *rows_end, const int32_t *col_indx, const double *values);
mi_sparse_status_t mi_sparse_c_create_bsr(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                          int64_t cols, int64_t block_size, const int32_t *rows_start,
                                          const int32_t *rows_end, const int32_t *col_indx,
                                          const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_bsr(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                          int64_t cols, int64_t block_size, const int32_t *rows_start,
                                          const int32_t *rows_end, const int32_t *col_indx,
                                          const mi_complex16 *values);
mi_sparse_status_t mi_sparse_s_create_bsr_64(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                             int64_t cols, int64_t block_size, const int64_t *rows_start,
                                             const int64_t *rows_end, const int64_t *col_indx,
                                             const float *values);
mi_sparse_status_t mi_sparse_d_create_bsr_64(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                             int64_t cols, int64_t block_size, const int64_t *rows_start,
                                             const int64_t *rows_end, const int64_t *col_indx,
                                             const double *values);
mi_sparse_status_t mi_sparse_c_create_bsr_64(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                             int64_t cols, int64_t block_size, const int64_t *rows_start,
                                             const int64_t *rows_end, const int64_t *col_indx,
                                             const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_create_bsr_64(mi_sparse_matrix_t *A, int base, int block_layout, int64_t rows,
                                             int64_t cols, int64_t block_size, const int64_t *rows_start,
                                             const int64_t *rows_end, const int64_t *col_indx,
                                             const mi_complex16 *values);

/* mkl_sparse_destroy (reference _cfunctions.py:447-452; _common.py:671-680).  Frees every
 * library-owned buffer of the handle; never frees caller memory.  NULL -> NOT_INITIALIZED. */
mi_sparse_status_t mi_sparse_destroy(mi_sparse_matrix_t A);

/* mkl_sparse_order (reference _cfunctions.py:447-452; _common.py:683-692): sort the column
 * indices inside every row (values permuted alike, stable for duplicates).  Like MKL this
 * re-orders the storage the handle aliases: for a handle created from HOST arrays the sorted
 * indices / values are also written back to the caller's arrays (the reference documents that
 * inputs "may be reordered in place", README.md:45-46). */
mi_sparse_status_t mi_sparse_order(mi_sparse_matrix_t A);
/* mkl_sparse_optimize analogue (MKL's inspector stage; the reference never calls it -- it creates a handle per product,
 * _common.py:245-293 -- so nothing binds it there).  For a handle that will be the left operand of several mi_sparse_?_mm
 * products: builds now what the library otherwise builds behind its first three products (row partition, fix-up schedule,
 * and for >= 2^21 entries the column-partitioned form of the long rows), so the next product already runs the steady-state
 * kernels.  Costs one pass over the matrix (~2-3 ms at 3e7 entries) and a second copy of its entries. */
mi_sparse_status_t mi_sparse_optimize(mi_sparse_matrix_t A);

/* mkl_sparse_convert_csr (reference call site _common.py:705-707): new CSR handle holding
 * op(A) converted from whatever format A was created in.  op must be 10 (the reference never
 * passes anything else). */
mi_sparse_status_t mi_sparse_convert_csr(mi_sparse_matrix_t A, int op, mi_sparse_matrix_t *out);

/* mkl_sparse_?_export_csr (reference _cfunctions.py:554-564; _common.py:452-461).  Returns HOST
 * pointers to library-owned copies that stay valid until the handle is destroyed (the reference
 * copies them out before destroy, _common.py:488-491).  rows_end == rows_start + 1 always.
 * The 32-bit flavour returns ALLOC_FAILED when nnz or a dimension exceeds INT32_MAX (the same
 * status MKL LP64 gives; the reference appends its ILP64 hint to it, _common.py:658-659). */
mi_sparse_status_t mi_sparse_s_export_csr(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **rows_start, int32_t **rows_end, int32_t **col_indx,
                                          float **values);
mi_sparse_status_t mi_sparse_d_export_csr(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **rows_start, int32_t **rows_end, int32_t **col_indx,
                                          double **values);
mi_sparse_status_t mi_sparse_c_export_csr(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **rows_start, int32_t **rows_end, int32_t **col_indx,
                                          mi_complex8 **values);
mi_sparse_status_t mi_sparse_z_export_csr(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **rows_start, int32_t **rows_end, int32_t **col_indx,
                                          mi_complex16 **values);
mi_sparse_status_t mi_sparse_s_export_csr_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **rows_start, int64_t **rows_end, int64_t **col_indx,
                                             float **values);
mi_sparse_status_t mi_sparse_d_export_csr_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **rows_start, int64_t **rows_end, int64_t **col_indx,
                                             double **values);
mi_sparse_status_t mi_sparse_c_export_csr_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **rows_start, int64_t **rows_end, int64_t **col_indx,
                                             mi_complex8 **values);
mi_sparse_status_t mi_sparse_z_export_csr_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **rows_start, int64_t **rows_end, int64_t **col_indx,
                                             mi_complex16 **values);

/* mkl_sparse_?_export_csc (reference _cfunctions.py:554-564): the CSC arrays of the handle
 * (cols_start / cols_end / row_indx); converts on the device if the handle holds CSR. */
mi_sparse_status_t mi_sparse_s_export_csc(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **cols_start, int32_t **cols_end, int32_t **row_indx,
                                          float **values);
mi_sparse_status_t mi_sparse_d_export_csc(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **cols_start, int32_t **cols_end, int32_t **row_indx,
                                          double **values);
mi_sparse_status_t mi_sparse_c_export_csc(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **cols_start, int32_t **cols_end, int32_t **row_indx,
                                          mi_complex8 **values);
mi_sparse_status_t mi_sparse_z_export_csc(mi_sparse_matrix_t A, int *base, int32_t *rows, int32_t *cols,
                                          int32_t **cols_start, int32_t **cols_end, int32_t **row_indx,
                                          mi_complex16 **values);
mi_sparse_status_t mi_sparse_s_export_csc_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **cols_start, int64_t **cols_end, int64_t **row_indx,
                                             float **values);
mi_sparse_status_t mi_sparse_d_export_csc_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **cols_start, int64_t **cols_end, int64_t **row_indx,
                                             double **values);
mi_sparse_status_t mi_sparse_c_export_csc_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **cols_start, int64_t **cols_end, int64_t **row_indx,
                                             mi_complex8 **values);
mi_sparse_status_t mi_sparse_z_export_csc_64(mi_sparse_matrix_t A, int *base, int64_t *rows, int64_t *cols,
                                             int64_t **cols_start, int64_t **cols_end, int64_t **row_indx,
                                             mi_complex16 **values);

/* Build-specific helpers with no MKL analogue ------------------------------------------------ */

/* Shape, entry count, value type ('s','d','c','z') and native index width (4 or 8) of a handle.
 * Any output pointer may be NULL. */
mi_sparse_status_t mi_sparse_get_info(mi_sparse_matrix_t A, int64_t *rows, int64_t *cols, int64_t *nnz,
                                      char *value_type, int *index_bytes);

/* DEVICE pointers of the handle's canonical CSR arrays -- indptr: int64_t[rows + 1], col_indx:
 * int32_t[nnz], values: nnz elements of the handle's value type -- valid until the handle is
 * destroyed or re-ordered (mi_sparse_order may move a library-owned result's values to a new
 * block: fetch the pointers again after it).  Lets HBM-resident callers consume spmm / syrk results
 * without a host round trip. */
/* mkl_sparse_?_export_bsr (reference _cfunctions.py:567-579; call site _common.py:540-551): the handle's matrix
 * re-blocked with the block size it was created with (create_bsr) or produced with (mi_sparse_spmm of two BSR handles
 * of one block size); blocks row-major (*block_layout = 101), library-owned host arrays valid until destroy.
 * A handle created from BSR arrays and not ordered since exports those arrays unchanged (block order, block layout,
 * explicit zeros -- what MKL's aliasing handle gives back; reference tests/test_mkl.py:230-249).
 * rows / cols / block_size are in BLOCKS.  Handles without a block size: NOT_SUPPORTED. */
mi_sparse_status_t mi_sparse_s_export_bsr(mi_sparse_matrix_t A, int *base, int *block_layout, int32_t *rows,
                                            int32_t *cols, int32_t *block_size, int32_t **rows_start,
                                            int32_t **rows_end, int32_t **col_indx, float **values);
mi_sparse_status_t mi_sparse_s_export_bsr_64(mi_sparse_matrix_t A, int *base, int *block_layout, int64_t *rows,
                                               int64_t *cols, int64_t *block_size, int64_t **rows_start,
                                               int64_t **rows_end, int64_t **col_indx, float **values);
mi_sparse_status_t mi_sparse_d_export_bsr(mi_sparse_matrix_t A, int *base, int *block_layout, int32_t *rows,
                                            int32_t *cols, int32_t *block_size, int32_t **rows_start,
                                            int32_t **rows_end, int32_t **col_indx, double **values);
mi_sparse_status_t mi_sparse_d_export_bsr_64(mi_sparse_matrix_t A, int *base, int *block_layout, int64_t *rows,
                                               int64_t *cols, int64_t *block_size, int64_t **rows_start,
                                               int64_t **rows_end, int64_t **col_indx, double **values);
mi_sparse_status_t mi_sparse_c_export_bsr(mi_sparse_matrix_t A, int *base, int *block_layout, int32_t *rows,
                                            int32_t *cols, int32_t *block_size, int32_t **rows_start,
                                            int32_t **rows_end, int32_t **col_indx, mi_complex8 **values);
mi_sparse_status_t mi_sparse_c_export_bsr_64(mi_sparse_matrix_t A, int *base, int *block_layout, int64_t *rows,
                                               int64_t *cols, int64_t *block_size, int64_t **rows_start,
                                               int64_t **rows_end, int64_t **col_indx, mi_complex8 **values);
mi_sparse_status_t mi_sparse_z_export_bsr(mi_sparse_matrix_t A, int *base, int *block_layout, int32_t *rows,
                                            int32_t *cols, int32_t *block_size, int32_t **rows_start,
                                            int32_t **rows_end, int32_t **col_indx, mi_complex16 **values);
mi_sparse_status_t mi_sparse_z_export_bsr_64(mi_sparse_matrix_t A, int *base, int *block_layout, int64_t *rows,
                                               int64_t *cols, int64_t *block_size, int64_t **rows_start,
                                               int64_t **rows_end, int64_t **col_indx, mi_complex16 **values);

/* Copy the handle's matrix straight into CALLER-allocated arrays (host or device): indptr of rows + 1 (csc: cols + 1)
 * entries, indices and values of nnz entries (mi_sparse_get_info), indices of `index_bytes` (4 / 8) bytes each.  The
 * MKL-shaped mi_sparse_?_export_* calls hand out library-owned copies which the reference's Python then copies again
 * (reference _common.py:488-491); this is the one-copy form the package's own export uses. */
mi_sparse_status_t mi_sparse_copy_out(mi_sparse_matrix_t A, int csc, int index_bytes, void *indptr, void *indices,
                                      void *values);
mi_sparse_status_t mi_sparse_get_device_csr(mi_sparse_matrix_t A, void **indptr, void **col_indx,
                                            void **values);

/* ============================================================================================
 * Executor routines
 * ========================================================================================== */

/* mkl_sparse_?_mm (reference _cfunctions.py:612-625; call site _sparse_dense.py:111-123)
 *     C := alpha * op(A) * B + beta * C
 * A: sparse handle (any creation format; CSC/BSR are converted once and cached on the handle).
 * B: dense, op(A).cols x columns;  C: dense, op(A).rows x columns;  both `layout` (101 row
 * major, 102 column major) with leading dimensions ldb / ldc.  beta == 0 never reads C. */
mi_sparse_status_t mi_sparse_s_mm(int op, float alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  int layout, const float *B, int64_t columns, int64_t ldb, float beta,
                                  float *C, int64_t ldc);
mi_sparse_status_t mi_sparse_d_mm(int op, double alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  int layout, const double *B, int64_t columns, int64_t ldb, double beta,
                                  double *C, int64_t ldc);
mi_sparse_status_t mi_sparse_c_mm(int op, mi_complex8 alpha, mi_sparse_matrix_t A,
                                  struct mi_matrix_descr descr, int layout, const mi_complex8 *B,
                                  int64_t columns, int64_t ldb, mi_complex8 beta, mi_complex8 *C, int64_t ldc);
mi_sparse_status_t mi_sparse_z_mm(int op, mi_complex16 alpha, mi_sparse_matrix_t A,
                                  struct mi_matrix_descr descr, int layout, const mi_complex16 *B,
                                  int64_t columns, int64_t ldb, mi_complex16 beta, mi_complex16 *C,
                                  int64_t ldc);

/* mkl_sparse_?_mv (reference _cfunctions.py:628-637; call site _sparse_vector.py:87-95)
 *     y := alpha * op(A) * x + beta * y */
mi_sparse_status_t mi_sparse_s_mv(int op, float alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const float *x, float beta, float *y);
mi_sparse_status_t mi_sparse_d_mv(int op, double alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const double *x, double beta, double *y);
mi_sparse_status_t mi_sparse_c_mv(int op, mi_complex8 alpha, mi_sparse_matrix_t A,
                                  struct mi_matrix_descr descr, const mi_complex8 *x, mi_complex8 beta,
                                  mi_complex8 *y);
mi_sparse_status_t mi_sparse_z_mv(int op, mi_complex16 alpha, mi_sparse_matrix_t A,
                                  struct mi_matrix_descr descr, const mi_complex16 *x, mi_complex16 beta,
                                  mi_complex16 *y);

/* mkl_sparse_spmm (reference _cfunctions.py:376-382; call site _sparse_sparse.py:35-40)
 *     C := op(A) * B, sparse CSR, library-owned (free with mi_sparse_destroy(*C)).
 * Column indices inside a row are NOT ordered (call mi_sparse_order for that); entries whose
 * value cancels to 0.0 are kept, as MKL keeps them.  A and B must hold the same value type.
 * op must be 10 (the only value the reference passes). */
mi_sparse_status_t mi_sparse_spmm(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, mi_sparse_matrix_t *C);

/* The same product with the column indices of every row in INCREASING order: what the reference's
 * `reorder_output=True` obtains with mkl_sparse_spmm followed by mkl_sparse_order
 * (_sparse_sparse.py:226-230, _common.py:683-692) in one call.  Knowing that the rows will be ordered,
 * the library accumulates the long rows by the rank of their column (they come out in order: nothing
 * left to sort but the short rows) whenever the row bitmaps that takes fit in 8 GiB; otherwise it is
 * mi_sparse_spmm + mi_sparse_order.  Same pattern, same values (to tolerance) either way. */
mi_sparse_status_t mi_sparse_spmm_ordered(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, mi_sparse_matrix_t *C);

/* mkl_sparse_sp2m (the two-stage API; MKL signature
 *   mkl_sparse_sp2m(opA, descrA, A, opB, descrB, B, request, *C)):   C := op(A) * op(B)   with the symbolic and the
 * numeric phase of the two-phase hash SpGEMM callable separately, so that a pattern is analysed once and reused:
 *   request = NNZ_COUNT (91)              symbolic phase: *C is created, its row pointer and nnz are final
 *             FINALIZE_MULT (92)          numeric phase on the *C of an earlier NNZ_COUNT: column indices + values;
 *                                         may be repeated after mi_sparse_?_set_values on A / B (same patterns)
 *             FINALIZE_MULT_NO_VAL (94)   as 92 (the indices come out of the same pass as the values here)
 *             FULL_MULT (90), FULL_MULT_NO_VAL (93)   both phases, == mi_sparse_spmm for op = 10
 * op = 10 / 11 (12 for real types); descriptors must be GENERAL.  A and B must be the same handles in every stage. */
mi_sparse_status_t mi_sparse_sp2m(int op_a, struct mi_matrix_descr descr_a, mi_sparse_matrix_t A, int op_b,
                                  struct mi_matrix_descr descr_b, mi_sparse_matrix_t B, int request,
                                  mi_sparse_matrix_t *C);
/* mkl_sparse_sypr (reference _sparse_sypr.py:87-135, dead upstream):  C := triu(op(A) * B * op(A)^T), B symmetric,
 * given by its upper triangle (descr_b = {21, 41, 50}; entries below the diagonal are ignored), real types,
 * request = FULL_MULT (90).  C is a new handle (sparse CSR, rows unsorted). */
mi_sparse_status_t mi_sparse_sypr(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, struct mi_matrix_descr descr_b,
                                  mi_sparse_matrix_t *C, int request);
/* mkl_sparse_?_syprd (reference _sparse_sypr.py:29-84):  C := alpha * op(A) * B * op(A)^T + beta * C with a DENSE
 * symmetric B (upper triangle referenced) and dense C; only the upper triangle of C is read or written. */
mi_sparse_status_t mi_sparse_s_syprd(int op, mi_sparse_matrix_t A, const float *B, int layout_b, int64_t ldb,
                                     float alpha, float beta, float *C, int layout_c, int64_t ldc);
mi_sparse_status_t mi_sparse_d_syprd(int op, mi_sparse_matrix_t A, const double *B, int layout_b, int64_t ldb,
                                     double alpha, double beta, double *C, int layout_c, int64_t ldc);
/* Replace ALL values of a handle (storage order of the arrays it was created from; host or device pointer), keeping
 * its pattern and its plans -- the analogue of updating the aliased value array in place under MKL
 * (mkl_sparse_?_update_values): what makes NNZ_COUNT once / FINALIZE_MULT many times useful.
 * A handle created from DEVICE arrays aliases them, but it also keeps derived copies of the values (the cached
 * transpose, the packed records of the dense gram): values of aliased device arrays must therefore be changed through
 * this call, not by writing the arrays in place -- an in-place write leaves those copies stale.
 * The STRUCTURE (row pointer, column indices) of aliased device arrays must not be modified at all while the handle lives:
 * plans, sortedness answers and row-length bounds are cached on the handle for good. */
mi_sparse_status_t mi_sparse_s_set_values(mi_sparse_matrix_t A, const float *values);
mi_sparse_status_t mi_sparse_d_set_values(mi_sparse_matrix_t A, const double *values);
mi_sparse_status_t mi_sparse_c_set_values(mi_sparse_matrix_t A, const mi_complex8 *values);
mi_sparse_status_t mi_sparse_z_set_values(mi_sparse_matrix_t A, const mi_complex16 *values);

/* mkl_sparse_?_spmmd (reference _cfunctions.py:601-609; call site _sparse_sparse.py:94-101)
 *     dense C := op(A) * B (C is overwritten; there is no beta). */
mi_sparse_status_t mi_sparse_s_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, float *C,
                                     int64_t ldc);
mi_sparse_status_t mi_sparse_d_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, double *C,
                                     int64_t ldc);
mi_sparse_status_t mi_sparse_c_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout,
                                     mi_complex8 *C, int64_t ldc);
mi_sparse_status_t mi_sparse_z_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout,
                                     mi_complex16 *C, int64_t ldc);

/* mkl_sparse_syrk (reference _cfunctions.py:456-461; call site _gram_matrix.py:70-74)
 *     op = 10: C := A * A^T ;  op = 11 (or 12): C := A^T * A      (reference _gram_matrix.py:35-40)
 * C is the UPPER triangle (col >= row) as sparse CSR, library-owned, columns unordered. */
mi_sparse_status_t mi_sparse_syrk(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t *C);

/* mkl_sparse_?_syrkd (reference _cfunctions.py:640-649; call site _gram_matrix.py:149-157)
 *     op = 10: C := alpha * A * A^T + beta * C ;  op = 11: C := alpha * A^T * A + beta * C
 * Only the upper triangle (col >= row) of the dense n x n C is read or written; the strict
 * lower triangle is left untouched (MKL leaves it undefined). */
mi_sparse_status_t mi_sparse_s_syrkd(int op, mi_sparse_matrix_t A, float alpha, float beta, float *C,
                                     int layout, int64_t ldc);
mi_sparse_status_t mi_sparse_d_syrkd(int op, mi_sparse_matrix_t A, double alpha, double beta, double *C,
                                     int layout, int64_t ldc);
/* Row band of the same product (no MKL counterpart; the multi-GPU split of the gram path, SURVEY section 8e:
 * "split by output rows ... needs no reduction"): only output rows [row0, row1) are produced, and C points at the
 * (row1 - row0) x n block that holds them (row-major: ldc >= n; column-major: ldc >= row1 - row0).  Entries left of
 * the diagonal of the full matrix (column < row) are not touched.  mi_sparse_?_syrkd == rows [0, n). */
mi_sparse_status_t mi_sparse_s_syrkd_rows(int op, mi_sparse_matrix_t A, float alpha, float beta, float *C,
                                          int layout, int64_t ldc, int64_t row0, int64_t row1);
mi_sparse_status_t mi_sparse_d_syrkd_rows(int op, mi_sparse_matrix_t A, double alpha, double beta, double *C,
                                          int layout, int64_t ldc, int64_t row0, int64_t row1);

/* cblas_?gemm (reference _cfunctions.py:582-598; call site _dense_dense.py:53-66)
 *     C := alpha * op(A) * op(B) + beta * C     (the MFMA consumer; fallback path, test-sized)
 * c / z take alpha and beta BY POINTER, as CBLAS does. */
mi_sparse_status_t mi_cblas_sgemm(int layout, int transa, int transb, int64_t m, int64_t n, int64_t k,
                                  float alpha, const float *A, int64_t lda, const float *B, int64_t ldb,
                                  float beta, float *C, int64_t ldc);
mi_sparse_status_t mi_cblas_dgemm(int layout, int transa, int transb, int64_t m, int64_t n, int64_t k,
                                  double alpha, const double *A, int64_t lda, const double *B, int64_t ldb,
                                  double beta, double *C, int64_t ldc);
mi_sparse_status_t mi_cblas_cgemm(int layout, int transa, int transb, int64_t m, int64_t n, int64_t k,
                                  const mi_complex8 *alpha, const mi_complex8 *A, int64_t lda,
                                  const mi_complex8 *B, int64_t ldb, const mi_complex8 *beta, mi_complex8 *C,
                                  int64_t ldc);
mi_sparse_status_t mi_cblas_zgemm(int layout, int transa, int transb, int64_t m, int64_t n, int64_t k,
                                  const mi_complex16 *alpha, const mi_complex16 *A, int64_t lda,
                                  const mi_complex16 *B, int64_t ldb, const mi_complex16 *beta,
                                  mi_complex16 *C, int64_t ldc);

/* cblas_?syrk (reference _cfunctions.py:652-665; call site _gram_matrix.py:235-247)
 *     trans = 111: C := alpha * A * A^T + beta * C (A n x k); 112: C := alpha * A^T * A + beta * C
 * Only the `uplo` triangle of C is touched. */
mi_sparse_status_t mi_cblas_ssyrk(int layout, int uplo, int trans, int64_t n, int64_t k, float alpha,
                                  const float *A, int64_t lda, float beta, float *C, int64_t ldc);
mi_sparse_status_t mi_cblas_dsyrk(int layout, int uplo, int trans, int64_t n, int64_t k, double alpha,
                                  const double *A, int64_t lda, double beta, double *C, int64_t ldc);

/* ============================================================================================
 * Service routines (analogues of MKL_Get_Version_String / MKL_Get_Max_Threads /
 * MKL_Set_Interface_Layer, reference _cfunctions.py:729-771)
 * ========================================================================================== */

/* NUL-terminated version / device description written into buf (truncated to len). */
mi_sparse_status_t mi_sparse_get_version_string(char *buf, int len);
/* Number of visible HIP devices (0 when there is none; never fails). */
int mi_sparse_get_device_count(void);
/* Select the device used by the calling host thread for subsequent calls.  A thread that never
 * calls this works on the HIP device that is current on it at its first library call. */
mi_sparse_status_t mi_sparse_set_device(int device);
/* Device the calling thread's library context is bound to (-1: not bound yet). */
int mi_sparse_get_device(void);
/* Stream (a hipStream_t passed as void*) on which the calling thread's work is enqueued. */
mi_sparse_status_t mi_sparse_set_stream(void *hip_stream);
/* The stream set by mi_sparse_set_stream (NULL = the default stream), so that a caller can restore it. */
mi_sparse_status_t mi_sparse_get_stream(void **hip_stream);
/* Block until everything enqueued by the calling thread's stream has finished. */
mi_sparse_status_t mi_sparse_synchronize(void);
/* Reason for the calling thread's most recent non-zero status ("" if none). */
const char *mi_sparse_last_error(void);
/* Tuning / diagnostic knob: name -> integer value; unknown names return INVALID_VALUE.  Used by
 * bench.py / tools to A/B kernel variants; defaults are the shipped choice.  Names:
 *   spmm_chunk, spmm_unroll, spmm_hot_kb, spmm_hot_force, spmm_force_generic,
 *   spmm_slices (XCD-affine column slices: 0 = by row width, 1 / 2 / 4 / 8), spmm_plan_sync
 *                                                                               (SpMM kernel variants)
 *   spgemm_lds_parts, spgemm_slice_table, spgemm_slice_table_max, spgemm_part_log2s_bias,
 *   spgemm_force_global, spgemm_global_mode, spgemm_rank (1: big rows accumulate by rank on the stored row bitmaps and come
 *                   out sorted, 0 (default): range-partitioned LDS hash)                              (SpGEMM big-row paths)
 *   spgemm_hub (0 (default): off; 2: hub rows of a full product -- rows beyond the LDS hash classes -- through dense LDS
 *                   accumulators over popularity-ordered column blocks of a relabelled copy of B (csrc/spgemm_hub.inc), 1: the same
 *                   from spgemm_hub_min_products products on; 3: the relabelling alone.  Measured at parity with the default
 *                   range path on the literal configs[2]; tuning: spgemm_hub_fill_pct, spgemm_hub_acc_kb, spgemm_hub_block_kb)
 *   spgemm_onepass (1: a product whose rows all have <= 512 products runs as ONE kernel -- no symbolic pass, the rows of the
 *                   result are placed by a decoupled look-back; 0: always symbolic + numeric)
 *   transpose_lds_hist (1: the column histogram of a device transpose -- CSC operands, gram matrices -- runs through LDS ranges of
 *                   32768 counters when the matrix has >= 2^22 entries and <= 2^20 columns; 0: one global atomic per entry)
 *   spgemm_narrow_ptr (1: the upper-bound pass gathers B's row extents from an int32 copy of its row pointer made per call when
 *                   nnz(B) < 2^31; 0: from the int64 pointer)
 *   gram_sliced (1: slice-table walk when the slices are short, 2: whenever the rows are sorted, 0: never),
 *   gram_heads (1: slice bounds travel with the entries of X^T when rows have <= 255 entries; 0: per-row table),
 *   gram_tile_kb (0: 152 KiB tiles where they save a tile per output row, else 128; 64 / 128 / 152 force), gram_persistent (-1 auto, 0: one workgroup per tile, k: k workgroups per LDS slot),
 *   gram_queue (1: the sliced walk's (row, tile) pairs are pulled in order from one counter per XCD; 0: fixed stride per workgroup),
 *   bsr_native (0: BSR handles multiply through their CSR expansion), staged_copies (0: plain hipMemcpy for
 *   pageable host arrays)
 *   pool_enable (0: hipFree released device blocks at once), pool_max_mb (cap on cached bytes,
 *   -1 = half of the device memory), pool_trim (any value: return the cache to the driver now)
 *   deterministic (1: SpGEMM, sparse gram and dense gram return the same BITS on every run -- SpMM / SpMV always do.
 *                  The default kernels add the products of an entry with LDS atomics in arrival order; this mode orders the
 *                  result's columns and re-forms every value in a fixed order (one wave per row; dense gram: one wave per
 *                  tile).  A validation mode: several times slower, minutes on hub rows of > 1e8 products)
 *   profile_events, trace_phases                                               (diagnostics) */
mi_sparse_status_t mi_sparse_set_option(const char *name, int64_t value);
/* The device's copy rate (read + write bytes per second, GB/s) as this library's own tuned copy kernel reaches it: 16 bytes
 * per lane, 4 - 8 independent accesses in flight, plain and non-temporal, 4 - 32 workgroups per CU; the best variant over
 * `reps` timed launches each (hipEvents) on two fresh buffers of `bytes` bytes.  The "measured HBM roofline" of bench.py. */
mi_sparse_status_t mi_sparse_probe_copy(int64_t bytes, int reps, double *best_gbs);
/* Diagnostic counters of the calling thread.  With option "profile_events" = 1 the SpMM executor
 * brackets its main kernel with hipEvents on the launch stream and accumulates
 * "spmm_kernel_ms" (sum of durations) and "spmm_kernel_launches"; "reset" (any value pointer)
 * zeroes them.  Unknown names return INVALID_VALUE. */
mi_sparse_status_t mi_sparse_get_counter(const char *name, double *value);
/* Name of the dominant kernel the calling thread launched last, as the library instantiated it -- e.g.
 * "mi::k_spmm<float, V=4, LPN=16, U=4, TAG=1> x 2 column slices" -- so that measurement tools report what ran rather
 * than what they expect to have run.  Empty string before the first launch. */
mi_sparse_status_t mi_sparse_get_last_kernel(char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* MI_SPARSE_H */
