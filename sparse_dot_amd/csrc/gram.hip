// gram.hip -- dense-output gram product of a sparse matrix.
// Replaces mkl_sparse_?_syrkd (reference sparse_dot_mkl/_gram_matrix.py:149-157):
//     op = 10 :  C := alpha * A   * A^T + beta * C      (n = rows of A)
//     op = 11 :  C := alpha * A^T * A   + beta * C      (n = cols of A)
// Only the upper triangle (col >= row) of C is read or written.
//
// Row-owned formulation (no inter-workgroup races): with X = A (op 11) or X = A^T (op 10),
// C = X^T X and output row i is  sum over nonzeros (r, i) of X of  X[r, i] * X[r, i:].
// The CSR of X^T (cached on the handle) lists those nonzeros; one wave owns output row i, walks
// them in order and lets its lanes span the entries of X's row r.  The scatter into the dense
// row uses L2 float/double atomics (entries of different r collide on the same column); all
// traffic to one output row comes from one wave, so the row stays in that XCD's L2.
#include "common.hpp"

namespace mi {

template <typename T>
__global__ void __launch_bounds__(256)
    k_scale_upper(T* C, int64_t n, int64_t c_rs, int64_t c_cs, T beta, int beta_zero)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const bool row_major = (c_cs == 1);
    const int64_t i = row_major ? t / n : t % n;
    const int64_t j = row_major ? t % n : t / n;
    if (j < i) return;
    T* c = C + i * c_rs + j * c_cs;
    *c = beta_zero ? vt<T>::zero() : vt<T>::mul(beta, *c);
}

template <typename T>
__global__ void __launch_bounds__(256)
    k_syrkd(int64_t n, const int64_t* __restrict__ tptr, const int32_t* __restrict__ tcol,
            const T* __restrict__ tval, const int64_t* __restrict__ xptr, const int32_t* __restrict__ xcol,
            const T* __restrict__ xval, T* __restrict__ C, int64_t c_rs, int64_t c_cs, T alpha)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (i >= n) return;
    T* crow = C + i * c_rs;
    for (int64_t p = tptr[i]; p < tptr[i + 1]; ++p) {
        const int32_t r = tcol[p];
        const T a = vt<T>::mul(alpha, tval[p]);
        for (int64_t q = xptr[r] + lane; q < xptr[r + 1]; q += WAVE) {
            const int32_t j = xcol[q];
            if (j >= i) atomic_accum(crow + (int64_t)j * c_cs, vt<T>::mul(a, xval[q]));
        }
    }
}

template <typename T>
static int syrkd_generic(int op, mi_sparse_matrix_t A, T alpha, T beta, T* C, int layout, int64_t ldc)
{
    return guarded([&] {
        mi_sparse_matrix* h = check_handle(A);
        if (h->vtype != type_char<T>::value)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "handle holds '%c' values but the '%c' routine was called", h->vtype,
                 type_char<T>::value);
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE && op != MI_SPARSE_OPERATION_TRANSPOSE &&
            op != MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad operation code %d", op);
        if (layout != MI_SPARSE_LAYOUT_ROW_MAJOR && layout != MI_SPARSE_LAYOUT_COLUMN_MAJOR)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad layout code %d", layout);
        const bool aat = (op == MI_SPARSE_OPERATION_NON_TRANSPOSE);
        const int64_t n = aat ? h->rows : h->cols;
        if (n == 0) return;
        if (!C) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output array");
        if (ldc < n) fail(MI_SPARSE_STATUS_INVALID_VALUE, "ldc too small");
        Context& c = ctx();
        c.scratch_reset();
        // X = A (A^T A) or A^T (A A^T);  t = CSR of X^T, x = CSR of X
        Csr& x = aat ? need_csrT(h) : need_csr(h);
        Csr& t = aat ? need_csr(h) : need_csrT(h);
        const bool row_major = layout == MI_SPARSE_LAYOUT_ROW_MAJOR;
        const int64_t c_rs = row_major ? ldc : 1, c_cs = row_major ? 1 : ldc;
        Staged sc;
        const int beta_zero = vt<T>::is_zero(beta) ? 1 : 0;
        sc.stage_in(C, sizeof(T) * (size_t)((n - 1) * ldc + n), true);  // lower triangle must survive the round trip
        T* dC = static_cast<T*>(sc.dev);
        MI_LAUNCH((k_scale_upper<T>), dim3((unsigned)ceil_div(n * n, 256)), dim3(256), c.stream, dC, n, c_rs, c_cs, beta,
                  beta_zero);
        MI_LAUNCH((k_syrkd<T>), dim3((unsigned)ceil_div(n * WAVE, 256)), dim3(256), c.stream, n, (const int64_t*)t.ptr,
                  (const int32_t*)t.col, (const T*)t.val, (const int64_t*)x.ptr, (const int32_t*)x.col,
                  (const T*)x.val, dC, c_rs, c_cs, alpha);
        MI_HIP_CHECK(hipGetLastError());
        sc.copy_back();
    });
}

}  // namespace mi

extern "C" {

mi_sparse_status_t mi_sparse_s_syrkd(int op, mi_sparse_matrix_t A, float alpha, float beta, float* C, int layout,
                                     int64_t ldc)
{
    return mi::syrkd_generic<float>(op, A, alpha, beta, C, layout, ldc);
}
mi_sparse_status_t mi_sparse_d_syrkd(int op, mi_sparse_matrix_t A, double alpha, double beta, double* C, int layout,
                                     int64_t ldc)
{
    return mi::syrkd_generic<double>(op, A, alpha, beta, C, layout, ldc);
}

}  // extern "C"
