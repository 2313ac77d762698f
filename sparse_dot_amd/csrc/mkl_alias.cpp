// mkl_alias.cpp -> libmi_mkl_rt.so: the MKL-named face of libmi_sparse.so.
//
// "a replacement .so selected via $MKL_RT would have to export ..." -- this thin library exports
// exactly the symbols the reference binds at import (reference sparse_dot_mkl/_mkl_interface/_cfunctions.py:43-168)
// under MKL's own names and argument conventions and forwards the hot-path ones to the mi_* C ABI
// (include/mi_sparse.h), so that the UNMODIFIED reference package runs on this backend with
//     MKL_RT=/path/to/libmi_mkl_rt.so  python -c "import sparse_dot_mkl"
// Integer width: MKL_INT follows MKL_Set_Interface_Layer (0 = LP64 / 32-bit, the default; 1 = ILP64 / 64-bit), as in
// MKL.  Every integer argument is declared 64 bits wide here and truncated under LP64: on the x86-64 SysV ABI an
// `int` argument travels in a full register / stack slot whose upper half is unspecified, so one symbol serves both
// layers.  Index ARRAYS are read as int32 or int64 accordingly (mi_*_create_* / *_64).
// Out of the hot path (QR solver, PARDISO, CG / FGMRES): present so that the import
// resolves, they return SPARSE_STATUS_NOT_SUPPORTED / an error code and do nothing.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/mi_sparse.h"

typedef long long mkl_int;  // see the header comment
typedef mi_sparse_matrix_t H;

static int g_ilp64 = 0;
static inline int64_t I(mkl_int x) { return g_ilp64 ? (int64_t)x : (int64_t)(int32_t)x; }

extern "C" {

// ---- service ------------------------------------------------------------------------------------
int MKL_Set_Interface_Layer(int code)
{
    if (code == 0 || code == 1) g_ilp64 = code;
    return g_ilp64;
}
int MKL_Get_Max_Threads(void) { return 1; }          // the device is the parallelism; there is no host thread pool
void MKL_Set_Num_Threads(int) {}
int MKL_Set_Num_Threads_Local(int) { return 0; }
void mkl_free_buffers(void) { (void)mi_sparse_set_option("pool_trim", 1); }

struct MKLVersion {
    int MajorVersion, MinorVersion, UpdateVersion;
    char* ProductStatus;
    char* Build;
    char* Processor;
    char* Platform;
};
void MKL_Get_Version(MKLVersion* v)
{
    static char status[128] = "mi_sparse (MI355X backend behind the MKL names)", build[128] = "0.1.0",
                proc[128] = "AMD Instinct MI355X (gfx950)", plat[128] = "ROCm / HIP";
    if (!v) return;
    v->MajorVersion = 2024;  // the reference only warns below 2020 (__init__.py:160-163)
    v->MinorVersion = 0;
    v->UpdateVersion = 0;
    v->ProductStatus = status;
    v->Build = build;
    v->Processor = proc;
    v->Platform = plat;
}
void MKL_Get_Version_String(char* buf, int len)
{
    if (buf && len > 0 && mi_sparse_get_version_string(buf, len) != 0) buf[0] = 0;
}

// ---- handles --------------------------------------------------------------------------------------
#define ALIAS_CREATE(t, CT)                                                                                              \
    int mkl_sparse_##t##_create_csr(H* A, int base, mkl_int rows, mkl_int cols, void* rs, void* re, void* ci, CT* v)     \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_create_csr_64(A, base, I(rows), I(cols), (const int64_t*)rs, (const int64_t*)re, \
                                                       (const int64_t*)ci, v)                                            \
                       : mi_sparse_##t##_create_csr(A, base, I(rows), I(cols), (const int32_t*)rs, (const int32_t*)re,   \
                                                    (const int32_t*)ci, v);                                              \
    }                                                                                                                    \
    int mkl_sparse_##t##_create_csc(H* A, int base, mkl_int rows, mkl_int cols, void* cs, void* ce, void* ri, CT* v)     \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_create_csc_64(A, base, I(rows), I(cols), (const int64_t*)cs, (const int64_t*)ce, \
                                                       (const int64_t*)ri, v)                                            \
                       : mi_sparse_##t##_create_csc(A, base, I(rows), I(cols), (const int32_t*)cs, (const int32_t*)ce,   \
                                                    (const int32_t*)ri, v);                                              \
    }                                                                                                                    \
    int mkl_sparse_##t##_create_bsr(H* A, int base, int layout, mkl_int rows, mkl_int cols, mkl_int bs, void* rs,        \
                                    void* re, void* ci, CT* v)                                                           \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_create_bsr_64(A, base, layout, I(rows), I(cols), I(bs), (const int64_t*)rs,     \
                                                       (const int64_t*)re, (const int64_t*)ci, v)                        \
                       : mi_sparse_##t##_create_bsr(A, base, layout, I(rows), I(cols), I(bs), (const int32_t*)rs,        \
                                                    (const int32_t*)re, (const int32_t*)ci, v);                          \
    }                                                                                                                    \
    int mkl_sparse_##t##_export_csr(H A, int* base, void* rows, void* cols, void** rs, void** re, void** ci, CT** v)     \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_export_csr_64(A, base, (int64_t*)rows, (int64_t*)cols, (int64_t**)rs,           \
                                                       (int64_t**)re, (int64_t**)ci, v)                                  \
                       : mi_sparse_##t##_export_csr(A, base, (int32_t*)rows, (int32_t*)cols, (int32_t**)rs,              \
                                                    (int32_t**)re, (int32_t**)ci, v);                                    \
    }                                                                                                                    \
    int mkl_sparse_##t##_export_csc(H A, int* base, void* rows, void* cols, void** cs, void** ce, void** ri, CT** v)     \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_export_csc_64(A, base, (int64_t*)rows, (int64_t*)cols, (int64_t**)cs,           \
                                                       (int64_t**)ce, (int64_t**)ri, v)                                  \
                       : mi_sparse_##t##_export_csc(A, base, (int32_t*)rows, (int32_t*)cols, (int32_t**)cs,              \
                                                    (int32_t**)ce, (int32_t**)ri, v);                                    \
    }                                                                                                                    \
    int mkl_sparse_##t##_export_bsr(H A, int* base, int* layout, void* rows, void* cols, void* bs, void** rs, void** re, \
                                    void** ci, CT** v)                                                                   \
    {                                                                                                                    \
        return g_ilp64 ? mi_sparse_##t##_export_bsr_64(A, base, layout, (int64_t*)rows, (int64_t*)cols, (int64_t*)bs,     \
                                                       (int64_t**)rs, (int64_t**)re, (int64_t**)ci, v)                   \
                       : mi_sparse_##t##_export_bsr(A, base, layout, (int32_t*)rows, (int32_t*)cols, (int32_t*)bs,        \
                                                    (int32_t**)rs, (int32_t**)re, (int32_t**)ci, v);                     \
    }
ALIAS_CREATE(s, float)
ALIAS_CREATE(d, double)
ALIAS_CREATE(c, mi_complex8)
ALIAS_CREATE(z, mi_complex16)

int mkl_sparse_destroy(H A) { return mi_sparse_destroy(A); }
int mkl_sparse_order(H A) { return mi_sparse_order(A); }
int mkl_sparse_convert_csr(H A, int op, H* out) { return mi_sparse_convert_csr(A, op, out); }

// ---- executors --------------------------------------------------------------------------------------
#define ALIAS_EXEC(t, CT)                                                                                                \
    int mkl_sparse_##t##_mm(int op, CT alpha, H A, struct mi_matrix_descr d, int layout, const CT* B, mkl_int n,          \
                            mkl_int ldb, CT beta, CT* C, mkl_int ldc)                                                    \
    {                                                                                                                    \
        return mi_sparse_##t##_mm(op, alpha, A, d, layout, B, I(n), I(ldb), beta, C, I(ldc));                            \
    }                                                                                                                    \
    int mkl_sparse_##t##_mv(int op, CT alpha, H A, struct mi_matrix_descr d, const CT* x, CT beta, CT* y)                 \
    {                                                                                                                    \
        return mi_sparse_##t##_mv(op, alpha, A, d, x, beta, y);                                                          \
    }                                                                                                                    \
    int mkl_sparse_##t##_spmmd(int op, H A, H B, int layout, CT* C, mkl_int ldc)                                         \
    {                                                                                                                    \
        return mi_sparse_##t##_spmmd(op, A, B, layout, C, I(ldc));                                                       \
    }
ALIAS_EXEC(s, float)
ALIAS_EXEC(d, double)
ALIAS_EXEC(c, mi_complex8)
ALIAS_EXEC(z, mi_complex16)

int mkl_sparse_spmm(int op, H A, H B, H* C) { return mi_sparse_spmm(op, A, B, C); }
int mkl_sparse_syrk(int op, H A, H* C) { return mi_sparse_syrk(op, A, C); }
int mkl_sparse_sp2m(int opa, struct mi_matrix_descr da, H A, int opb, struct mi_matrix_descr db, H B, int request, H* C)
{
    return mi_sparse_sp2m(opa, da, A, opb, db, B, request, C);
}
int mkl_sparse_sypr(int op, H A, H B, struct mi_matrix_descr db, H* C, int request)
{
    return mi_sparse_sypr(op, A, B, db, C, request);
}
int mkl_sparse_s_syrkd(int op, H A, float alpha, float beta, float* C, int layout, mkl_int ldc)
{
    return mi_sparse_s_syrkd(op, A, alpha, beta, C, layout, I(ldc));
}
int mkl_sparse_d_syrkd(int op, H A, double alpha, double beta, double* C, int layout, mkl_int ldc)
{
    return mi_sparse_d_syrkd(op, A, alpha, beta, C, layout, I(ldc));
}
// the reference rejects complex gram matrices before it gets here (_gram_matrix.py:296-299)
int mkl_sparse_c_syrkd(int, H, mi_complex8, mi_complex8, mi_complex8*, int, mkl_int) { return MI_SPARSE_STATUS_NOT_SUPPORTED; }
int mkl_sparse_z_syrkd(int, H, mi_complex16, mi_complex16, mi_complex16*, int, mkl_int) { return MI_SPARSE_STATUS_NOT_SUPPORTED; }
int mkl_sparse_s_syprd(int op, H A, const float* B, int lb, mkl_int ldb, float alpha, float beta, float* C, int lc, mkl_int ldc)
{
    return mi_sparse_s_syprd(op, A, B, lb, I(ldb), alpha, beta, C, lc, I(ldc));
}
int mkl_sparse_d_syprd(int op, H A, const double* B, int lb, mkl_int ldb, double alpha, double beta, double* C, int lc,
                       mkl_int ldc)
{
    return mi_sparse_d_syprd(op, A, B, lb, I(ldb), alpha, beta, C, lc, I(ldc));
}

// ---- CBLAS --------------------------------------------------------------------------------------------
void cblas_sgemm(int layout, int ta, int tb, mkl_int m, mkl_int n, mkl_int k, float alpha, const float* A, mkl_int lda,
                 const float* B, mkl_int ldb, float beta, float* C, mkl_int ldc)
{
    (void)mi_cblas_sgemm(layout, ta, tb, I(m), I(n), I(k), alpha, A, I(lda), B, I(ldb), beta, C, I(ldc));
}
void cblas_dgemm(int layout, int ta, int tb, mkl_int m, mkl_int n, mkl_int k, double alpha, const double* A, mkl_int lda,
                 const double* B, mkl_int ldb, double beta, double* C, mkl_int ldc)
{
    (void)mi_cblas_dgemm(layout, ta, tb, I(m), I(n), I(k), alpha, A, I(lda), B, I(ldb), beta, C, I(ldc));
}
void cblas_cgemm(int layout, int ta, int tb, mkl_int m, mkl_int n, mkl_int k, const void* alpha, const void* A,
                 mkl_int lda, const void* B, mkl_int ldb, const void* beta, void* C, mkl_int ldc)
{
    (void)mi_cblas_cgemm(layout, ta, tb, I(m), I(n), I(k), (const mi_complex8*)alpha, (const mi_complex8*)A, I(lda),
                         (const mi_complex8*)B, I(ldb), (const mi_complex8*)beta, (mi_complex8*)C, I(ldc));
}
void cblas_zgemm(int layout, int ta, int tb, mkl_int m, mkl_int n, mkl_int k, const void* alpha, const void* A,
                 mkl_int lda, const void* B, mkl_int ldb, const void* beta, void* C, mkl_int ldc)
{
    (void)mi_cblas_zgemm(layout, ta, tb, I(m), I(n), I(k), (const mi_complex16*)alpha, (const mi_complex16*)A, I(lda),
                         (const mi_complex16*)B, I(ldb), (const mi_complex16*)beta, (mi_complex16*)C, I(ldc));
}
void cblas_ssyrk(int layout, int uplo, int trans, mkl_int n, mkl_int k, float alpha, const float* A, mkl_int lda, float beta,
                 float* C, mkl_int ldc)
{
    (void)mi_cblas_ssyrk(layout, uplo, trans, I(n), I(k), alpha, A, I(lda), beta, C, I(ldc));
}
void cblas_dsyrk(int layout, int uplo, int trans, mkl_int n, mkl_int k, double alpha, const double* A, mkl_int lda,
                 double beta, double* C, mkl_int ldc)
{
    (void)mi_cblas_dsyrk(layout, uplo, trans, I(n), I(k), alpha, A, I(lda), beta, C, I(ldc));
}
void cblas_csyrk(void) {}  // unreachable in the reference (complex gram is rejected in Python)
void cblas_zsyrk(void) {}

// ---- bound at import by the reference, outside the hot path (OUT OF SCOPE) -------------
#define ALIAS_STUB(name) int name(void) { return MI_SPARSE_STATUS_NOT_SUPPORTED; }
ALIAS_STUB(mkl_sparse_qr_reorder)
ALIAS_STUB(mkl_sparse_s_qr_factorize)
ALIAS_STUB(mkl_sparse_d_qr_factorize)
ALIAS_STUB(mkl_sparse_s_qr_solve)
ALIAS_STUB(mkl_sparse_d_qr_solve)
ALIAS_STUB(pardisoinit)
ALIAS_STUB(pardiso)
ALIAS_STUB(dcg_init)
ALIAS_STUB(dcg_check)
ALIAS_STUB(dcg)
ALIAS_STUB(dcg_get)
ALIAS_STUB(dcgmrhs_init)
ALIAS_STUB(dcgmrhs_check)
ALIAS_STUB(dcgmrhs)
ALIAS_STUB(dcgmrhs_get)
ALIAS_STUB(dfgmres_init)
ALIAS_STUB(dfgmres_check)
ALIAS_STUB(dfgmres)
ALIAS_STUB(dfgmres_get)

}  // extern "C"
