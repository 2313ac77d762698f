// bsr.hip -- native block-sparse (BSR) x dense product: C := alpha * A * B + beta * C with A in BSR.
// Replaces mkl_sparse_?_mm on handles made by mkl_sparse_?_create_bsr (reference _common.py:327-384; the
// reference sends scipy bsr_matrix operands straight to MKL, _sparse_dense.py:78-81 converts them only for
// column-major dense operands).
//
// Why a block kernel at all: the CSR expansion of a BSR matrix carries one 4-byte column index per VALUE; the block
// form carries one per bs x bs block and its gather unit is bs consecutive rows of B (bs * N * sizeof(T) contiguous
// bytes instead of N * sizeof(T)), so index traffic drops by bs^2 and every gather moves bs times more useful bytes.
// Why not MFMA (SURVEY section 8 f3 suggests it): the product is bound by that gather, not by arithmetic, and on
// gfx950 the f32-input MFMA runs at exactly the vector FMA rate (MI355X_MICROARCH.md, Matrix cores) while a
// bs = 4 block fills a quarter of the smallest 16x16x4 tile -- a VALU kernel with the block in registers does the
// same flops without the padding.
//
// One wave per block row.  LPN lanes x V values (16 bytes per lane) span the dense columns; the 64 / LPN lane groups
// take different blocks of the block row and are combined with xor-shuffles at the end.  Per block a lane group
// reads the block's bs x bs values (broadcast loads: every lane the same addresses) and bs coalesced 16 * LPN-byte
// segments of B, and does bs * bs * V FMAs per lane into bs accumulator rows.  Each output row is written exactly
// once (no atomics); empty block rows write beta * C.
#include "common.hpp"

namespace mi {

template <typename T>
__device__ __forceinline__ T bsr_shfl_xor(T v, int mask)
{
    return __shfl_xor(v, mask);
}
template <typename R>
__device__ __forceinline__ cx<R> bsr_shfl_xor(cx<R> v, int mask)
{
    return cx<R>{__shfl_xor(v.re, mask), __shfl_xor(v.im, mask)};
}

template <typename T, int V, int LPN, int BS>
__global__ void __launch_bounds__(256)
    k_bsr_spmm(int64_t brows, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol,
               const T* __restrict__ bval, int block_row_major, const T* __restrict__ B, int64_t b_rs,
               T* __restrict__ C, int64_t c_rs, int64_t N, T alpha, T beta, int beta_zero)
{
    constexpr int NG = WAVE / LPN;
    const int64_t bi = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;  // block row of this wave
    if (bi >= brows) return;
    const int lane = threadIdx.x % WAVE;
    const int g = lane / LPN, li = lane % LPN;
    const int64_t p0 = bptr[bi], p1 = bptr[bi + 1];
    for (int64_t j0 = 0; j0 < N; j0 += (int64_t)LPN * V) {
        const int64_t jc = j0 + (int64_t)li * V;
        const bool col_ok = jc < N;
        const T* bcolp = B + (col_ok ? jc : 0);
        T acc[BS][V];
#pragma unroll
        for (int r = 0; r < BS; ++r)
#pragma unroll
            for (int v = 0; v < V; ++v) acc[r][v] = vt<T>::zero();
        for (int64_t p = p0 + g; p < p1; p += NG) {
            const int64_t kb = bcol[p];
            const T* blk = bval + p * (BS * BS);
            vec<T, V> brow[BS];
#pragma unroll
            for (int kk = 0; kk < BS; ++kk) brow[kk] = *reinterpret_cast<const vec<T, V>*>(bcolp + (kb * BS + kk) * b_rs);
#pragma unroll
            for (int r = 0; r < BS; ++r) {
#pragma unroll
                for (int kk = 0; kk < BS; ++kk) {
                    const T a = block_row_major ? blk[r * BS + kk] : blk[kk * BS + r];
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[r][v] = vt<T>::fma(a, brow[kk].v[v], acc[r][v]);
                }
            }
        }
#pragma unroll
        for (int off = LPN; off < WAVE; off <<= 1)
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int v = 0; v < V; ++v) acc[r][v] = vt<T>::add(acc[r][v], bsr_shfl_xor(acc[r][v], off));
        if (g == 0 && col_ok) {
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                T* crow = C + (bi * BS + r) * c_rs + jc;
                vec<T, V> out;
                if (beta_zero) {
#pragma unroll
                    for (int v = 0; v < V; ++v) out.v[v] = vt<T>::mul(alpha, acc[r][v]);
                } else {
                    const vec<T, V> old = *reinterpret_cast<const vec<T, V>*>(crow);
#pragma unroll
                    for (int v = 0; v < V; ++v) out.v[v] = vt<T>::fma(alpha, acc[r][v], vt<T>::mul(beta, old.v[v]));
                }
                *reinterpret_cast<vec<T, V>*>(crow) = out;
            }
        }
    }
}

template <typename T, int V, int LPN>
static void launch_bsr(const Bsr& b, const T* B, int64_t ldb, T* C, int64_t ldc, int64_t N, T alpha, T beta)
{
    Context& c = ctx();
    const dim3 grid((unsigned)ceil_div(b.brows * WAVE, 256));
    const int rm = b.layout == MI_SPARSE_LAYOUT_ROW_MAJOR ? 1 : 0;
    const int bz = vt<T>::is_zero(beta) ? 1 : 0;
#define MI_BSR_LAUNCH(BSV)                                                                                          \
    MI_LAUNCH((k_bsr_spmm<T, V, LPN, BSV>), grid, dim3(256), c.stream, b.brows, (const int64_t*)b.ptr,               \
              (const int32_t*)b.col, (const T*)b.val, rm, B, ldb, C, ldc, N, alpha, beta, bz)
    switch (b.bs) {
        case 2: MI_BSR_LAUNCH(2); break;
        case 4: MI_BSR_LAUNCH(4); break;
        case 8: MI_BSR_LAUNCH(8); break;
        default: fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "no block kernel for block size %lld", (long long)b.bs);
    }
#undef MI_BSR_LAUNCH
}

// Can this call run on the block kernel?  (row-major dense operands on the 16-byte vector path, block sizes with a
// register-resident block; everything else goes through the handle's CSR expansion.)
template <typename T>
bool bsr_spmm_applicable(const Bsr& b, int layout, const T* B, int64_t N, int64_t ldb, const T* C, int64_t ldc)
{
    constexpr int V16 = 16 / (int)sizeof(T);
    if (!b.valid || options().bsr_native == 0) return false;
    if (b.bs != 2 && b.bs != 4 && !(b.bs == 8 && sizeof(T) <= 8)) return false;
    if (layout != MI_SPARSE_LAYOUT_ROW_MAJOR || N < 1 || N % V16 != 0) return false;
    if ((ldb * (int64_t)sizeof(T)) % 16 || (ldc * (int64_t)sizeof(T)) % 16) return false;
    if (reinterpret_cast<uintptr_t>(B) % 16 || reinterpret_cast<uintptr_t>(C) % 16) return false;
    return true;
}

template <typename T>
void bsr_spmm_device(const Bsr& b, T alpha, const T* B, int64_t N, int64_t ldb, T beta, T* C, int64_t ldc)
{
    if (b.brows == 0 || N == 0) return;
    constexpr int V16 = 16 / (int)sizeof(T);
    const int64_t lanes = N / V16;
    if (lanes >= 64) launch_bsr<T, V16, 64>(b, B, ldb, C, ldc, N, alpha, beta);
    else if (lanes > 16) launch_bsr<T, V16, 32>(b, B, ldb, C, ldc, N, alpha, beta);
    else if (lanes > 8) launch_bsr<T, V16, 16>(b, B, ldb, C, ldc, N, alpha, beta);
    else launch_bsr<T, V16, 8>(b, B, ldb, C, ldc, N, alpha, beta);
    counters().bsr_native_calls += 1.0;
}

#define MI_BSR_INST(T)                                                                                                  \
    template bool bsr_spmm_applicable<T>(const Bsr&, int, const T*, int64_t, int64_t, const T*, int64_t);                \
    template void bsr_spmm_device<T>(const Bsr&, T, const T*, int64_t, int64_t, T, T*, int64_t);
MI_BSR_INST(float)
MI_BSR_INST(double)
MI_BSR_INST(cfloat)
MI_BSR_INST(cdouble)
#undef MI_BSR_INST

}  // namespace mi
