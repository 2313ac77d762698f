// dense.hip -- dense x dense fallback: C := alpha * op(A) * op(B) + beta * C and the one-triangle
// SYRK.  Replaces cblas_?gemm (reference sparse_dot_mkl/_dense_dense.py:53-66) and cblas_?syrk
// (reference _gram_matrix.py:235-247).  This is the ONLY matrix-core (MFMA) consumer of the
// library; it is a correctness-first fallback for test-sized operands, not a tuned GEMM.
//
// Real types: 64 x 64 output tile per 256-thread workgroup, K stepped by 16 through LDS; each
// of the 4 waves owns a 32 x 32 quadrant:
//   float  : one v_mfma_f32_32x32x2_f32 accumulator (16 VGPRs), 8 MFMAs per K step
//            A operand lane l -> A[i = l & 31][k = l >> 5], B operand -> B[k = l >> 5][j = l & 31],
//            D reg r of lane l -> row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col l & 31
//   double : 2 x 2 v_mfma_f64_16x16x4_f64 accumulators, 4 MFMAs each per K step
//            A -> A[l & 15][k = l >> 4], B -> B[k = l >> 4][l & 15], D reg r -> row (l >> 4) + 4 * r,
//            col l & 15
// Both are exact IEEE fma chains in k order (MI355X guide section 3).  Operands are addressed
// through (row stride, column stride) pairs so every layout / transpose combination runs the
// same kernel; tiles are zero-padded at the edges.  Complex types use a plain VALU kernel.
#include "common.hpp"

namespace mi {

constexpr int GT = 64;  // output tile
constexpr int GK = 16;  // K step

template <typename T>
__device__ __forceinline__ bool tri_keep(int tri, int64_t i, int64_t j)
{
    return tri == 0 || (tri == 1 ? j >= i : j <= i);
}

// tri: 0 full, 1 upper only, 2 lower only
template <typename T>
__global__ void __launch_bounds__(256)
    k_gemm_mfma(int64_t M, int64_t N, int64_t K, T alpha, const T* __restrict__ A, int64_t a_rs, int64_t a_cs,
                const T* __restrict__ B, int64_t b_rs, int64_t b_cs, T beta, int beta_zero, T* __restrict__ C,
                int64_t c_rs, int64_t c_cs, int tri)
{
    __shared__ T As[GT][GK + 1];  // [i][k]
    __shared__ T Bs[GK][GT + 1];  // [k][j]
    const int64_t i0 = (int64_t)blockIdx.y * GT, j0 = (int64_t)blockIdx.x * GT;
    if (tri == 1 && j0 + GT - 1 < i0) return;  // tile entirely below the diagonal
    if (tri == 2 && i0 + GT - 1 < j0) return;
    const int tid = threadIdx.x;
    const int wave = tid / WAVE, lane = tid % WAVE;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;  // quadrant origin inside the tile

    constexpr bool is_f32 = std::is_same<T, float>::value;
    f32x16 acc32 = {0};
    f64x4 acc64[2][2] = {{{0}, {0}}, {{0}, {0}}};

    for (int64_t k0 = 0; k0 < K; k0 += GK) {
        // stage A tile (64 x 16) and B tile (16 x 64), zero padded
        for (int e = tid; e < GT * GK; e += 256) {
            const int ai = e / GK, ak = e % GK;
            const int64_t gi = i0 + ai, gk = k0 + ak;
            As[ai][ak] = (gi < M && gk < K) ? A[gi * a_rs + gk * a_cs] : T(0);
            const int bk = e / GT, bj = e % GT;
            const int64_t gk2 = k0 + bk, gj = j0 + bj;
            Bs[bk][bj] = (gk2 < K && gj < N) ? B[gk2 * b_rs + gj * b_cs] : T(0);
        }
        __syncthreads();
        if constexpr (is_f32) {
#pragma unroll
            for (int kk = 0; kk < GK; kk += 2) {
                const float a = As[wi + (lane & 31)][kk + (lane >> 5)];
                const float b = Bs[kk + (lane >> 5)][wj + (lane & 31)];
                acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < GK; kk += 4) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const double a = As[wi + ti * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) {
                        const double b = Bs[kk + (lane >> 4)][wj + tj * 16 + (lane & 15)];
                        acc64[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc64[ti][tj], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    auto store = [&](int64_t gi, int64_t gj, T v) {
        if (gi < M && gj < N && tri_keep<T>(tri, gi, gj)) {
            T* c = C + gi * c_rs + gj * c_cs;
            *c = beta_zero ? alpha * v : alpha * v + beta * (*c);
        }
    };
    if constexpr (is_f32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            store(i0 + wi + row, j0 + wj + (lane & 31), acc32[r]);
        }
    } else {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store(i0 + wi + ti * 16 + (lane >> 4) + 4 * r, j0 + wj + tj * 16 + (lane & 15), acc64[ti][tj][r]);
    }
    (void)wave;
}

// complex (and any) types: one thread per output element
template <typename T>
__global__ void __launch_bounds__(256)
    k_gemm_valu(int64_t M, int64_t N, int64_t K, T alpha, const T* __restrict__ A, int64_t a_rs, int64_t a_cs, int conj_a,
                const T* __restrict__ B, int64_t b_rs, int64_t b_cs, int conj_b, T beta, int beta_zero,
                T* __restrict__ C, int64_t c_rs, int64_t c_cs, int tri)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * N) return;
    const bool row_major = (c_cs == 1);
    const int64_t i = row_major ? t / N : t % M;
    const int64_t j = row_major ? t % N : t / M;
    if (!(tri == 0 || (tri == 1 ? j >= i : j <= i))) return;
    T s = vt<T>::zero();
    for (int64_t k = 0; k < K; ++k) {
        T a = A[i * a_rs + k * a_cs];
        T b = B[k * b_rs + j * b_cs];
        if (conj_a) a = vt<T>::conj(a);
        if (conj_b) b = vt<T>::conj(b);
        s = vt<T>::fma(a, b, s);
    }
    T* c = C + i * c_rs + j * c_cs;
    *c = beta_zero ? vt<T>::mul(alpha, s) : vt<T>::fma(alpha, s, vt<T>::mul(beta, *c));
}

static size_t extent2(int64_t r, int64_t cdim, int64_t rs, int64_t cs)
{
    if (r == 0 || cdim == 0) return 0;
    return (size_t)((r - 1) * rs + (cdim - 1) * cs + 1);
}

// op(A): m x k, op(B): k x n
template <typename T>
static void gemm_run(int layout, int ta, int tb, int64_t m, int64_t n, int64_t k, T alpha, const T* A, int64_t lda,
                     const T* B, int64_t ldb, T beta, T* C, int64_t ldc, int tri)
{
    if (layout != MI_SPARSE_LAYOUT_ROW_MAJOR && layout != MI_SPARSE_LAYOUT_COLUMN_MAJOR)
        fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad layout code %d", layout);
    for (int t : {ta, tb})
        if (t != MI_CBLAS_NO_TRANS && t != MI_CBLAS_TRANS && t != MI_CBLAS_CONJ_TRANS)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad transpose code %d", t);
    if (m < 0 || n < 0 || k < 0) fail(MI_SPARSE_STATUS_INVALID_VALUE, "negative dimension");
    if (m == 0 || n == 0) return;
    if (!C || (k > 0 && (!A || !B))) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL dense operand");
    const bool rm = layout == MI_SPARSE_LAYOUT_ROW_MAJOR;
    // strides of op(X)(i, p): stored matrix is (rows x cols) with leading dimension ld
    auto strides = [&](int trans, int64_t ld, int64_t& rs, int64_t& cs) {
        const int64_t srs = rm ? ld : 1, scs = rm ? 1 : ld;  // strides of the stored matrix
        if (trans == MI_CBLAS_NO_TRANS) { rs = srs; cs = scs; } else { rs = scs; cs = srs; }
    };
    int64_t a_rs, a_cs, b_rs, b_cs;
    strides(ta, lda, a_rs, a_cs);
    strides(tb, ldb, b_rs, b_cs);
    const int64_t c_rs = rm ? ldc : 1, c_cs = rm ? 1 : ldc;
    Context& c = ctx();
    c.scratch_reset();
    Staged sa, sb, sc;
    sa.stage_in(A, sizeof(T) * extent2(m, k, a_rs, a_cs), true);
    sb.stage_in(B, sizeof(T) * extent2(k, n, b_rs, b_cs), true);
    sc.stage_in(C, sizeof(T) * extent2(m, n, c_rs, c_cs), true);
    const int beta_zero = vt<T>::is_zero(beta) ? 1 : 0;
    if constexpr (vt<T>::is_complex) {
        MI_LAUNCH((k_gemm_valu<T>), dim3((unsigned)ceil_div(m * n, 256)), dim3(256), c.stream, m, n, k, alpha,
                  static_cast<const T*>(sa.dev), a_rs, a_cs, (int)(ta == MI_CBLAS_CONJ_TRANS),
                  static_cast<const T*>(sb.dev), b_rs, b_cs, (int)(tb == MI_CBLAS_CONJ_TRANS), beta, beta_zero,
                  static_cast<T*>(sc.dev), c_rs, c_cs, tri);
    } else {
        MI_LAUNCH((k_gemm_mfma<T>), dim3((unsigned)ceil_div(n, GT), (unsigned)ceil_div(m, GT)), dim3(256), c.stream, m, n,
                  k, alpha, static_cast<const T*>(sa.dev), a_rs, a_cs, static_cast<const T*>(sb.dev), b_rs, b_cs, beta,
                  beta_zero, static_cast<T*>(sc.dev), c_rs, c_cs, tri);
    }
    MI_HIP_CHECK(hipGetLastError());
    if (sa.host || sb.host) c.sync();
    sc.copy_back();
}

}  // namespace mi

using mi::cdouble;
using mi::cfloat;

extern "C" {

mi_sparse_status_t mi_cblas_sgemm(int layout, int ta, int tb, int64_t m, int64_t n, int64_t k, float alpha,
                                  const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                                  int64_t ldc)
{
    return mi::guarded([&] { mi::gemm_run<float>(layout, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0); });
}
mi_sparse_status_t mi_cblas_dgemm(int layout, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha,
                                  const double* A, int64_t lda, const double* B, int64_t ldb, double beta, double* C,
                                  int64_t ldc)
{
    return mi::guarded([&] { mi::gemm_run<double>(layout, ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0); });
}
mi_sparse_status_t mi_cblas_cgemm(int layout, int ta, int tb, int64_t m, int64_t n, int64_t k,
                                  const mi_complex8* alpha, const mi_complex8* A, int64_t lda, const mi_complex8* B,
                                  int64_t ldb, const mi_complex8* beta, mi_complex8* C, int64_t ldc)
{
    return mi::guarded([&] {
        if (!alpha || !beta) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL scalar pointer");
        mi::gemm_run<cfloat>(layout, ta, tb, m, n, k, cfloat{alpha->real, alpha->imag}, (const cfloat*)A, lda,
                             (const cfloat*)B, ldb, cfloat{beta->real, beta->imag}, (cfloat*)C, ldc, 0);
    });
}
mi_sparse_status_t mi_cblas_zgemm(int layout, int ta, int tb, int64_t m, int64_t n, int64_t k,
                                  const mi_complex16* alpha, const mi_complex16* A, int64_t lda,
                                  const mi_complex16* B, int64_t ldb, const mi_complex16* beta, mi_complex16* C,
                                  int64_t ldc)
{
    return mi::guarded([&] {
        if (!alpha || !beta) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL scalar pointer");
        mi::gemm_run<cdouble>(layout, ta, tb, m, n, k, cdouble{alpha->real, alpha->imag}, (const cdouble*)A, lda,
                              (const cdouble*)B, ldb, cdouble{beta->real, beta->imag}, (cdouble*)C, ldc, 0);
    });
}

// syrk = gemm of A with its own transpose, one triangle stored
mi_sparse_status_t mi_cblas_ssyrk(int layout, int uplo, int trans, int64_t n, int64_t k, float alpha, const float* A,
                                  int64_t lda, float beta, float* C, int64_t ldc)
{
    return mi::guarded([&] {
        if (uplo != MI_CBLAS_UPPER && uplo != MI_CBLAS_LOWER) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad uplo");
        const int tri = uplo == MI_CBLAS_UPPER ? 1 : 2;
        if (trans == MI_CBLAS_NO_TRANS)
            mi::gemm_run<float>(layout, MI_CBLAS_NO_TRANS, MI_CBLAS_TRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri);
        else
            mi::gemm_run<float>(layout, MI_CBLAS_TRANS, MI_CBLAS_NO_TRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri);
    });
}
mi_sparse_status_t mi_cblas_dsyrk(int layout, int uplo, int trans, int64_t n, int64_t k, double alpha, const double* A,
                                  int64_t lda, double beta, double* C, int64_t ldc)
{
    return mi::guarded([&] {
        if (uplo != MI_CBLAS_UPPER && uplo != MI_CBLAS_LOWER) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad uplo");
        const int tri = uplo == MI_CBLAS_UPPER ? 1 : 2;
        if (trans == MI_CBLAS_NO_TRANS)
            mi::gemm_run<double>(layout, MI_CBLAS_NO_TRANS, MI_CBLAS_TRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri);
        else
            mi::gemm_run<double>(layout, MI_CBLAS_TRANS, MI_CBLAS_NO_TRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri);
    });
}

}  // extern "C"
