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

// ---- the same product for operands large enough to fill the chip (round 5): 128 x 128 tile per workgroup ------------------
// The 64 x 64 kernel above reaches 0.48 of the fp32 MFMA peak at 4096^3 (75.8 of 157.3 TFLOP/s; B stored transposed 0.33;
// fp64 39 of 78.6); this one 0.63-0.74 in fp32 over the four storage orders (98-116 TFLOP/s, K steps of 32), tools/gpu_gemm.py:
// its K step is a scalar gather of 8 elements per thread behind two barriers, with nothing in flight while the MFMAs run.
// Here: four waves x (64 x 64) = 2 x 2 accumulator tiles (fp32, 32x32x2) or eight waves x (32 x 64) = 2 x 4 (fp64, 16x16x4) per wave -- an A / B fragment
// read from LDS feeds two / four MFMAs --, K stepped by 32 (fp32) through TWO LDS buffers: the next step's panel of A and B is
// requested from HBM (16-byte loads along whichever dimension is contiguous, 4 / 8-byte ones otherwise) BEFORE this step's 32 /
// 64 MFMAs per wave are issued and written to the other buffer after them: one barrier per step, loads under the matrix pipe.
// LDS images are k-major (As[k][i], Bs[k][j]): a fragment read is 32 (16) consecutive words per k, conflict-free.
#ifndef MI_GEMM_BK32
#define MI_GEMM_BK32 32
#endif
constexpr int GB = 128;   // output tile
template <typename T>
constexpr int gemm_bk() { return sizeof(T) == 4 ? MI_GEMM_BK32 : 16; }   // K step
// row padding of the LDS images: fp32 fragments are 32 consecutive words of one row (conflict-free with any stride); an fp64
// fragment read takes 16 doubles from each of two k rows per half-wave: the row stride must be 32 words off a multiple of 64
template <typename T>
constexpr int gemm_pad() { return sizeof(T) == 8 ? 16 : 4; }

// how a panel is read from HBM: 0 element by element, 1 vectors along the tile's row / column index, 2 vectors along k
__host__ __device__ inline int gemm_vec_mode(int64_t s_mn, int64_t s_k, int64_t ld_ok16)
{
    if (!ld_ok16) return 0;
    if (s_mn == 1) return 1;
    if (s_k == 1) return 2;
    return 0;
}

template <typename T>
constexpr int gemm_threads() { return sizeof(T) == 4 ? 256 : 512; }

template <typename T>
__global__ void __launch_bounds__(gemm_threads<T>())
    k_gemm_mfma128(int64_t M, int64_t N, int64_t K, T alpha, const T* __restrict__ A, int64_t a_rs, int64_t a_cs, int a_mode,
                   const T* __restrict__ B, int64_t b_rs, int64_t b_cs, int b_mode, T beta, int beta_zero,
                   T* __restrict__ C, int64_t c_rs, int64_t c_cs, int tri, int tiles_n)
{
    constexpr bool is_f32 = std::is_same<T, float>::value;
    constexpr int GBK = gemm_bk<T>();
    constexpr int V = 16 / (int)sizeof(T);            // elements per 16-byte vector
    constexpr int NT = gemm_threads<T>();             // 256 (fp32: four waves x 64 x 64) / 512 (fp64: eight waves x 32 x 64)
    constexpr int EPT = GB * GBK / NT;                // elements of each panel per thread
    constexpr int VPT = EPT / V;                      // vectors of each panel per thread (2 / 4)
    constexpr int GBP = gemm_pad<T>();
    __shared__ __attribute__((aligned(16))) T As[2][GBK][GB + GBP];
    __shared__ __attribute__((aligned(16))) T Bs[2][GBK][GB + GBP];
    // tile order: groups of 8 tile rows walk the tile columns together (the A panels of a group and the B panel of a column
    // are shared by workgroups that run at the same time)
    const int64_t tiles_m = (M + GB - 1) / GB;
    int64_t bi, bj;
    {
        const int64_t b = blockIdx.x, grp = 8, per = grp * tiles_n;
        const int64_t g0 = (b / per) * grp, rows_here = tiles_m - g0 < grp ? tiles_m - g0 : grp;
        bi = g0 + (b % per) % rows_here;
        bj = (b % per) / rows_here;
    }
    const int64_t i0 = bi * GB, j0 = bj * GB;
    if (tri == 1 && j0 + GB - 1 < i0) return;  // tile entirely below the diagonal
    if (tri == 2 && i0 + GB - 1 < j0) return;
    const int tid = threadIdx.x, wave = tid / WAVE, lane = tid % WAVE;
    const int wi = is_f32 ? (wave >> 1) * 64 : (wave >> 1) * 32, wj = (wave & 1) * 64;

    // ---- HBM -> registers -> LDS, one panel of A (GB x GBK) and one of B (GBK x GB) per K step ----
    T ra[EPT], rb[EPT];
    auto fetch = [&](int64_t k0) {
        // A: element (i, k) of the panel
        if (a_mode == 1) {  // vectors along i (A stored with i contiguous)
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, kk = e / GB, ii = e % GB;
                const int64_t gi = i0 + ii, gk = k0 + kk;
                if (gi + V <= M && gk < K) {
                    const vec<T, V> x = *reinterpret_cast<const vec<T, V>*>(A + gi * a_rs + gk * a_cs);
#pragma unroll
                    for (int q = 0; q < V; ++q) ra[v * V + q] = x.v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) ra[v * V + q] = (gi + q < M && gk < K) ? A[(gi + q) * a_rs + gk * a_cs] : T(0);
                }
            }
        } else if (a_mode == 2) {  // vectors along k
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, ii = e / GBK, kk = e % GBK;
                const int64_t gi = i0 + ii, gk = k0 + kk;
                if (gi < M && gk + V <= K) {
                    const vec<T, V> x = *reinterpret_cast<const vec<T, V>*>(A + gi * a_rs + gk * a_cs);
#pragma unroll
                    for (int q = 0; q < V; ++q) ra[v * V + q] = x.v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) ra[v * V + q] = (gi < M && gk + q < K) ? A[gi * a_rs + (gk + q) * a_cs] : T(0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + u * NT, kk = e / GB, ii = e % GB;
                const int64_t gi = i0 + ii, gk = k0 + kk;
                ra[u] = (gi < M && gk < K) ? A[gi * a_rs + gk * a_cs] : T(0);
            }
        }
        // B: element (k, j) of the panel
        if (b_mode == 1) {  // vectors along j
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, kk = e / GB, jj = e % GB;
                const int64_t gj = j0 + jj, gk = k0 + kk;
                if (gj + V <= N && gk < K) {
                    const vec<T, V> x = *reinterpret_cast<const vec<T, V>*>(B + gk * b_rs + gj * b_cs);
#pragma unroll
                    for (int q = 0; q < V; ++q) rb[v * V + q] = x.v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) rb[v * V + q] = (gj + q < N && gk < K) ? B[gk * b_rs + (gj + q) * b_cs] : T(0);
                }
            }
        } else if (b_mode == 2) {  // vectors along k
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, jj = e / GBK, kk = e % GBK;
                const int64_t gj = j0 + jj, gk = k0 + kk;
                if (gj < N && gk + V <= K) {
                    const vec<T, V> x = *reinterpret_cast<const vec<T, V>*>(B + gk * b_rs + gj * b_cs);
#pragma unroll
                    for (int q = 0; q < V; ++q) rb[v * V + q] = x.v[q];
                } else {
#pragma unroll
                    for (int q = 0; q < V; ++q) rb[v * V + q] = (gj < N && gk + q < K) ? B[(gk + q) * b_rs + gj * b_cs] : T(0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + u * NT, kk = e / GB, jj = e % GB;
                const int64_t gj = j0 + jj, gk = k0 + kk;
                rb[u] = (gj < N && gk < K) ? B[gk * b_rs + gj * b_cs] : T(0);
            }
        }
    };
    auto stash = [&](int buf) {
        if (a_mode == 1) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, kk = e / GB, ii = e % GB;
                vec<T, V> x;
#pragma unroll
                for (int q = 0; q < V; ++q) x.v[q] = ra[v * V + q];
                *reinterpret_cast<vec<T, V>*>(&As[buf][kk][ii]) = x;
            }
        } else if (a_mode == 2) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, ii = e / GBK, kk = e % GBK;
#pragma unroll
                for (int q = 0; q < V; ++q) As[buf][kk + q][ii] = ra[v * V + q];
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + u * NT;
                As[buf][e / GB][e % GB] = ra[u];
            }
        }
        if (b_mode == 1) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, kk = e / GB, jj = e % GB;
                vec<T, V> x;
#pragma unroll
                for (int q = 0; q < V; ++q) x.v[q] = rb[v * V + q];
                *reinterpret_cast<vec<T, V>*>(&Bs[buf][kk][jj]) = x;
            }
        } else if (b_mode == 2) {
#pragma unroll
            for (int v = 0; v < VPT; ++v) {
                const int e = (tid + v * NT) * V, jj = e / GBK, kk = e % GBK;
#pragma unroll
                for (int q = 0; q < V; ++q) Bs[buf][kk + q][jj] = rb[v * V + q];
            }
        } else {
#pragma unroll
            for (int u = 0; u < EPT; ++u) {
                const int e = tid + u * NT;
                Bs[buf][e / GB][e % GB] = rb[u];
            }
        }
    };

    f32x16 acc32[2][2];
    f64x4 acc64[2][4];
    if constexpr (is_f32) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc32[a][b][r] = 0.f;
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc64[a][b][r] = 0.0;
    }
    // interior tiles with vector loads on both operands: the address of every load of a panel is formed ONCE (a pointer per
    // vector, advanced by one K step per iteration) -- the general `fetch` above spends ~300 instructions per step on index
    // arithmetic, 64-bit products and bounds, and two workgroups that share a CU run in lockstep (launched together, same
    // work), so that phase is not hidden behind the other workgroup's MFMAs: 0.59 -> MfmaUtil of the fp32 kernel at 4096^3
    const bool fast = a_mode != 0 && b_mode != 0 && i0 + GB <= M && j0 + GB <= N;
    const T* pa[VPT];
    const T* pb[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const int e = (tid + v * NT) * V;
        const int a_ii = a_mode == 1 ? e % GB : e / GBK, a_kk = a_mode == 1 ? e / GB : e % GBK;
        const int b_jj = b_mode == 1 ? e % GB : e / GBK, b_kk = b_mode == 1 ? e / GB : e % GBK;
        pa[v] = A + (i0 + a_ii) * a_rs + (int64_t)a_kk * a_cs;
        pb[v] = B + (int64_t)b_kk * b_rs + (j0 + b_jj) * b_cs;
    }
    const int64_t a_step = (int64_t)GBK * a_cs, b_step = (int64_t)GBK * b_rs;
    auto fetch_fast = [&]() {  // the next full K step of both panels
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            pa[v] += a_step;
            pb[v] += b_step;
            const vec<T, V> x = *reinterpret_cast<const vec<T, V>*>(pa[v]);
            const vec<T, V> y = *reinterpret_cast<const vec<T, V>*>(pb[v]);
#pragma unroll
            for (int q = 0; q < V; ++q) {
                ra[v * V + q] = x.v[q];
                rb[v * V + q] = y.v[q];
            }
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    int cur = 0;
    auto mma = [&](int cur) {
        if constexpr (is_f32) {
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 2) {
                const int kr = kk + (lane >> 5), c = lane & 31;
                const float a0 = As[cur][kr][wi + c], a1 = As[cur][kr][wi + 32 + c];
                const float b0 = Bs[cur][kr][wj + c], b1 = Bs[cur][kr][wj + 32 + c];
                acc32[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc32[0][0], 0, 0, 0);
                acc32[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc32[0][1], 0, 0, 0);
                acc32[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc32[1][0], 0, 0, 0);
                acc32[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc32[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < GBK; kk += 4) {
                const int kr = kk + (lane >> 4), c = lane & 15;
                double a[2], b[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < 2) a[t] = As[cur][kr][wi + t * 16 + c];
                    b[t] = Bs[cur][kr][wj + t * 16 + c];
                }
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
                        acc64[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], acc64[ti][tj], 0, 0, 0);
            }
        }
    };
    int64_t k0 = 0;
    if (fast) {  // full K steps of interior tiles: requests, MFMAs and LDS writes in one straight line
        for (; k0 + 2 * GBK <= K; k0 += GBK) {
            fetch_fast();
            mma(cur);
            stash(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    for (; k0 < K; k0 += GBK) {
        const bool more = k0 + GBK < K;
        if (more) fetch(k0 + GBK);  // in flight under this step's MFMAs
        mma(cur);
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    auto store = [&](int64_t gi, int64_t gj, T v) {
        if (gi < M && gj < N && tri_keep<T>(tri, gi, gj)) {
            T* c = C + gi * c_rs + gj * c_cs;
            *c = beta_zero ? alpha * v : alpha * v + beta * (*c);
        }
    };
    if constexpr (is_f32) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    store(i0 + wi + a * 32 + row, j0 + wj + b * 32 + (lane & 31), acc32[a][b][r]);
                }
    } else {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    store(i0 + wi + ti * 16 + (lane >> 4) + 4 * r, j0 + wj + tj * 16 + (lane & 15), acc64[ti][tj][r]);
    }
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
        // operands that fill the chip with 128 x 128 tiles take the pipelined kernel (k_gemm_mfma128), the rest the 64 x 64 one
        const int64_t tm = ceil_div(m, (int64_t)GB), tn = ceil_div(n, (int64_t)GB);
        const int64_t big_min = options().gemm_big_tiles >= 0 ? options().gemm_big_tiles : (int64_t)std::max(c.cus, 1);
        if (options().gemm_big_tiles != 0 && tm * tn >= big_min && k >= 2 * gemm_bk<T>() && tm * tn < ((int64_t)1 << 31)) {
            const T* ad = static_cast<const T*>(sa.dev);
            const T* bd = static_cast<const T*>(sb.dev);
            auto ok16 = [&](const T* ptr, int64_t ld) {  // 16-byte vectors: base and every row / column start aligned
                return (reinterpret_cast<uintptr_t>(ptr) % 16 == 0) && ((ld * (int64_t)sizeof(T)) % 16 == 0);
            };
            // the stored leading dimension is whichever of the two strides is not 1
            const int a_mode = gemm_vec_mode(a_rs, a_cs, ok16(ad, a_rs == 1 ? a_cs : a_rs));
            const int b_mode = gemm_vec_mode(b_cs, b_rs, ok16(bd, b_cs == 1 ? b_rs : b_cs));
            MI_LAUNCH((k_gemm_mfma128<T>), dim3((unsigned)(tm * tn)), dim3(gemm_threads<T>()), c.stream, m, n, k, alpha, ad, a_rs, a_cs, a_mode, bd,
                      b_rs, b_cs, b_mode, beta, beta_zero, static_cast<T*>(sc.dev), c_rs, c_cs, tri, (int)tn);
        } else {
            MI_LAUNCH((k_gemm_mfma<T>), dim3((unsigned)ceil_div(n, GT), (unsigned)ceil_div(m, GT)), dim3(256), c.stream, m, n,
                      k, alpha, static_cast<const T*>(sa.dev), a_rs, a_cs, static_cast<const T*>(sb.dev), b_rs, b_cs, beta,
                      beta_zero, static_cast<T*>(sc.dev), c_rs, c_cs, tri);
        }
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
