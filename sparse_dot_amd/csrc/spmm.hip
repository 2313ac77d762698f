// spmm.hip -- C := alpha * op(A) * B + beta * C   (CSR x dense), the primary hot path.
// Replaces mkl_sparse_?_mm / mkl_sparse_?_mv (reference sparse_dot_mkl/_sparse_dense.py:111-123,
// _sparse_vector.py:87-95).
//
// Kernel design (gfx950, 64-lane waves) -- DESIGN.md section 3.1
// -------------------------------------
// Work decomposition is nnz+row balanced ("merge path"): the sequence of all nonzeros with one
// extra "row end" item after every row has nnz + rows items; wave w takes items
// [w*CH, (w+1)*CH).  A row is OWNED by the wave that holds its row-end item; the owner writes the
// output row exactly once with plain stores (no atomics, no pre-zeroing of C: empty rows are
// just row-end items).  A row cut by a chunk boundary leaves a partial sum ("carry") in a small
// workspace (one N-vector per wave at most); a second tiny kernel adds the carries of a row, in
// chunk order, to the owner's output -> results are deterministic run to run.
//
// Inside a wave the lanes span the DENSE dimension: LPN lanes x V values (16 bytes per lane when
// the layout allows) cover one row of B, so every B-row read is a fully coalesced 16*LPN-byte
// segment; the 64/LPN lane groups take consecutive nonzeros.  The wave's slice of A is staged once
// in LDS and then streamed NG x U nonzeros at a time across row ends (round 5; rows one after the other before).  B rows go straight from L2/HBM to registers: a B row is used by exactly one
// nonzero of a workgroup, there is nothing to stage.
//
// Which rows of B an XCD's private 4 MB L2 holds is the lever (round 5): rows of A with many
// entries are split by part(column) into 8 sub-rows, partition p is multiplied on XCD p only
// (SpmmKpart / SpmmParts), the partial rows are combined in a fixed order (k_kp_combine).
#include <algorithm>

#include "common.hpp"

namespace mi {

template <typename T>
__device__ __forceinline__ T shfl_xor_val(T v, int mask)
{
    return __shfl_xor(v, mask);
}
template <typename R>
__device__ __forceinline__ cx<R> shfl_xor_val(cx<R> v, int mask)
{
    return cx<R>{__shfl_xor(v.re, mask), __shfl_xor(v.im, mask)};
}

// chunk_row[w] = the row that contains work item w*chunk (items of row r are
// [ptr[r] + r, ptr[r+1] + r]: its nonzeros, then its row-end item); `rows` past the last item
__global__ void k_spmm_plan(const int64_t* ptr, int64_t rows, int64_t nnz, int64_t chunk, int64_t nchunks,
                            int32_t* chunk_row)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > nchunks) return;
    const int64_t s = w * chunk;
    if (s >= nnz + rows) {
        chunk_row[w] = (int32_t)rows;
        return;
    }
    // smallest r in [0, rows] with start(r) = ptr[r] + r > s   (start is strictly increasing), minus one
    int64_t lo = 0, hi = rows;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (ptr[mid] + mid > s) hi = mid; else lo = mid + 1;
    }
    chunk_row[w] = (int32_t)(lo - 1);
}

constexpr int SPMM_WAVES = 4;  // waves per workgroup (independent of each other after staging)
// Rows of at most SPMM_SPLIT items are never cut: the wave that holds a short row's FIRST item
// processes the whole row (reading up to SPMM_SPLIT entries past its chunk) and the next wave skips
// it.  Only longer rows are cut at chunk boundaries and go through the carry / fix-up path.
constexpr int SPMM_SPLIT = 128;

// The long row (more than SPMM_SPLIT items) that chunk w cuts at its end and leaves a carry for, or
// -1.  Must agree with the staging logic of k_spmm.
__device__ __forceinline__ int32_t spmm_trail_row(const int64_t* ptr, const int32_t* chunk_row, int64_t rows,
                                                  int64_t nnz, int64_t ch, int64_t w)
{
    const int64_t total = nnz + rows;
    const int64_t s = w * ch;
    const int64_t e = (s + ch < total) ? s + ch : total;
    const int64_t rb = chunk_row[w + 1];
    if (rb < rows && (ptr[rb] + rb) < e && (ptr[rb + 1] - ptr[rb] + 1) > SPMM_SPLIT) return (int32_t)rb;
    return -1;
}

// Fix-up tasks with at most FIX_SMALL carries are summed by one lane group each, longer ones by a whole workgroup
// (k_spmm_fixup): the plan keeps them in two lists -- short tasks from the front of `tasks`, long ones from its back.
constexpr int FIX_SMALL = 8;

// one thread per chunk: the first chunk of every run of chunks that carry into the same row emits
// the fix-up task (row, first chunk, last chunk).  Tasks are independent, so their order in the lists
// (atomic cursors: n_tasks[0] short ones, n_tasks[1] long ones) does not matter; there are at most nchunks of them in all
// (runs are disjoint).
__global__ void k_spmm_plan_tasks(const int64_t* ptr, const int32_t* chunk_row, int64_t rows, int64_t nnz, int64_t ch,
                                  int64_t nchunks, int32_t* tasks, unsigned long long* n_tasks)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nchunks) return;
    const int32_t row = spmm_trail_row(ptr, chunk_row, rows, nnz, ch, w);
    if (row < 0) return;
    if (w > 0 && spmm_trail_row(ptr, chunk_row, rows, nnz, ch, w - 1) == row) return;
    int64_t last = w;
    while (last + 1 < nchunks && spmm_trail_row(ptr, chunk_row, rows, nnz, ch, last + 1) == row) ++last;
    int64_t t;
    if (last - w < FIX_SMALL) t = (int64_t)atomicAdd(n_tasks, 1ull);
    else t = nchunks - 1 - (int64_t)atomicAdd(n_tasks + 1, 1ull);
    tasks[3 * t + 0] = row;
    tasks[3 * t + 1] = (int32_t)w;
    tasks[3 * t + 2] = (int32_t)last;
}

// Everything a wave needs to know about its chunk, in ONE 32-byte load (round 3): used by k_spmv, whose chunks are short
// (one memory round trip of x gathers), so the chain  chunk_row[w], [w + 1] -> four row-pointer entries -> loads of A  was a
// visible share of a wave's life: 0.204-0.209 -> 0.194-0.202 ms (R-MAT), 0.245 -> 0.233 ms (uniform).  k_spmm keeps
// deriving it in the kernel: with the descriptor its time ROSE by 1 % (1.787-1.803 -> 1.810-1.829 ms, profiles/
// r03_spmm_chunk_desc_ab.log) -- the row-pointer lines it touches first are the ones its row-end loads need next.
// Same ownership rules as k_spmm's prologue, evaluated once in the plan.
struct alignas(32) SpmmChunk {
    int64_t P0, P1;    // the chunk's nonzeros: [P0, P1) in A's arrays
    int32_t r0;        // first row it processes
    int32_t nproc;     // rows it processes (the last one is a cut long row when has_trail)
    int32_t has_trail; // 1: the last processed row is cut at the chunk end and leaves a carry
    int32_t pad;
};

__global__ void k_spmm_plan_desc(const int64_t* __restrict__ ptr, const int32_t* __restrict__ chunk_row, int64_t rows,
                                 int64_t nnz, int64_t ch, int64_t nchunks, SpmmChunk* __restrict__ desc)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nchunks) return;
    const int64_t total = nnz + rows;
    const int64_t s = w * ch;
    const int64_t e = (s + ch < total) ? s + ch : total;
    const int64_t ra = chunk_row[w];      // row holding item s
    const int64_t rb = chunk_row[w + 1];  // row holding item e (== rows after the last item)
    int64_t r0, P0;
    {  // first row: a short row that began in the previous chunk was finished there; a long one is continued from item s
        const int64_t pa = ptr[ra], pa1 = ptr[ra + 1];
        const bool before = (pa + ra) < s;
        const bool is_long = (pa1 - pa + 1) > SPMM_SPLIT;
        if (before && !is_long) {
            r0 = ra + 1;
            P0 = pa1;
        } else {
            r0 = ra;
            P0 = before ? s - ra : pa;
        }
    }
    // last row: the row holding item e, if it starts inside this chunk, is finished here when short and cut (-> carry) when long
    int64_t r_stop, P1;
    int has_trail = 0;
    if (rb < rows && (ptr[rb] + rb) < e) {
        const int64_t pb = ptr[rb], pb1 = ptr[rb + 1];
        r_stop = rb + 1;
        if ((pb1 - pb + 1) > SPMM_SPLIT) {
            P1 = (e - rb < pb1) ? e - rb : pb1;
            has_trail = 1;
        } else {
            P1 = pb1;
        }
    } else {
        r_stop = rb;
        P1 = (rb < rows) ? ptr[rb] : nnz;
    }
    if (r_stop < r0) r_stop = r0;
    SpmmChunk d;
    d.P0 = P0;
    d.P1 = P1 > P0 ? P1 : P0;
    d.r0 = (int32_t)r0;
    d.nproc = (int32_t)(r_stop - r0);
    d.has_trail = has_trail;
    d.pad = 0;
    desc[w] = d;
}

// LDS bytes one wave needs for a chunk of CH items
template <typename T>
__host__ __device__ constexpr size_t spmm_wave_lds(int ch)
{
    return (size_t)(ch + SPMM_SPLIT) * sizeof(SpEntry<T>) + (size_t)(ch + 2) * sizeof(int32_t);
}

// TAG != 0: the staged column indices carry a "cold" flag in bit 31 (plan.col_tagged).  Hot rows of B are
// fetched with the default cache policy, cold rows with the non-temporal one (the cache policy is an
// immediate of the instruction, so the two flavours are two instructions selected per lane group), which
// keeps the streaming majority of the gather from evicting the few MB of B rows that power-law matrices
// hit over and over.
//   TAG == 1: raw buffer loads, 32-bit byte offsets (one VGPR of address per load) -- B below 4 GiB.  The
//             resource spans the whole 32-bit offset range (num_records = 0xffffffff): the range check
//             returns ZERO for any offset at or beyond num_records, so a smaller constant silently
//             drops rows of B (tests/test_gpu_baseline_configs.py::test_spmm_tagged_gather_large_dense_operand).
// Operands of 4 GiB and more (BASELINE configs[4]: B = 17 GB) gather UNTAGGED.  Tried in round 3 and dropped:
//   * structured buffer loads (index * stride formed by the hardware): gfx950 forms index * stride + offset in 32 bits,
//     rows of B beyond 4 GiB wrapped around (caught by test_config5_shape_single_gpu, profiles/r03_gpu_session_b.log);
//   * 64-bit global loads with the non-temporal hint on the cold ones: two global loads that differ only in the hint
//     are merged by the compiler and the hint is dropped -- through a second pointer argument, an under-aligned vector
//     type or a select of pointers alike (the policy is an immediate only on the buffer forms).
constexpr int SPMM_TAG_NONE = 0, SPMM_TAG_BUFFER = 1;

// Column-partitioned launch (SpmmKpart, common.hpp): the matrix is the concatenation of P sub-matrices, the chunks
// [cs[p], cs[p + 1]) of its plan belong to sub-matrix p, and sub-matrix p is processed by the XCDs x with x % P == p only
// (x / P picks one of the 8 / P column slices of the dense operand).  P == 0: the plain mapping.
// (Column slices IN TIME on top -- the chunks walked T times, one slice each -- were measured and removed in round 5:
// 1.44 -> 1.51 / 1.73 ms for T = 2 / 4, narrower requests cost more than the resident rows gain.)
struct SpmmParts {
    int64_t cs[9];
    int P;
};

// k_spmm -- the SpMM kernel.  Partition, ownership rules and carries as described at the top of the file; lane layout: LPN
// lanes x V values across the (slice of a) row of B, 64 / LPN lane groups on consecutive nonzeros.  The walk inside the wave
// (round 5; rounds 1-4 took the rows of a chunk one after the other -- a row of 7 nonzeros was one mostly empty batch of
// loads and one full memory round trip, a chunk of 30 short rows 30 dependent round trips where its 220 nonzeros fill 14
// batches: the short-row half of the column-partitioned product ran at 0.58 ms for a 0.32 ms gather): the wave streams
// through the chunk's staged nonzeros NG x U at a time WHATEVER the row structure: lane group g takes the nonzeros p + g,
// p + g + NG, ... of the stream, the row state (current row, its end) is wave-uniform -- scalar registers, scalar branches
// -- and a row end inside a batch combines the groups' accumulators (xor shuffles), writes the row (or the cut row's carry)
// and clears them.  The summation order is a function of the chunk alone: the same bits on every call.
template <typename T, int V>
__device__ __forceinline__ vec<T, V> spmm_tagged_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, bool cold)
{
    constexpr int BYTES = V * (int)sizeof(T);
    static_assert(BYTES == 16, "tagged gather: 16-byte pieces");
    u32x4 r;
    if (cold) r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2);  // nt
    else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
    return __builtin_bit_cast(vec<T, V>, r);
}

template <typename T, int V, int LPN, int U, int TAG>
__global__ void __launch_bounds__(SPMM_WAVES* WAVE, (U == 4 && sizeof(T) <= 8) ? 8 : 1)
    k_spmm(int64_t rows, int64_t nnz, const int64_t* __restrict__ ptr, const int32_t* __restrict__ col,
                const T* __restrict__ val, const int32_t* __restrict__ chunk_row, int64_t nchunks, int ch, int conj_a,
                const T* __restrict__ B, int64_t b_rs, int64_t b_cs, T* __restrict__ C, int64_t c_rs, int64_t c_cs,
                int64_t N, T alpha, T beta, int beta_zero, T* __restrict__ carry_val, int slices, SpmmParts parts)
{
    MI_DYN_SMEM(smem);
    constexpr int NG = WAVE / LPN;
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / WAVE));
    const int lane = threadIdx.x % WAVE;
    int64_t cb = blockIdx.x;
    int64_t jlo = 0, jhi = N;
    int64_t w;
    bool active;
    if (parts.P > 0) {  // column partitions: XCD x works on sub-matrix x % P, column slice x / P (slices == 8 / P)
        const int xcd = (int)(blockIdx.x & 7u);
        const int pp = xcd % parts.P;
        const int64_t ns = N / slices;
        jlo = (xcd / parts.P) * ns;
        jhi = jlo + ns;
        w = parts.cs[pp] + (int64_t)(blockIdx.x >> 3) * SPMM_WAVES + wave_in_block;
        active = w < parts.cs[pp + 1];
    } else {
        // XCD-affine column slicing: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only -- any
        // placement is correct).  With S slices the XCDs are split into S sets, set s only ever touches dense columns
        // [s * N / S, (s + 1) * N / S): its L2 holds N / S values per row of B instead of N, i.e. S times more hot rows.
        // Each chunk of A is then processed once per slice.
        if (slices > 1) {  // slices in {2, 4, 8}
            const int xcd = (int)(blockIdx.x & 7u);
            const int per = 8 / slices;
            cb = (int64_t)(blockIdx.x >> 3) * per + (xcd % per);
            const int64_t ns = N / slices;
            jlo = (xcd / per) * ns;
            jhi = jlo + ns;
        }
        w = cb * SPMM_WAVES + wave_in_block;
        active = w < nchunks;
    }
    const size_t per_wave = (spmm_wave_lds<T>(ch) + 15) & ~size_t(15);
    char* base = smem + per_wave * wave_in_block;
    SpEntry<T>* s_nz = reinterpret_cast<SpEntry<T>*>(base);
    int32_t* s_end = reinterpret_cast<int32_t*>(base + sizeof(SpEntry<T>) * (size_t)(ch + SPMM_SPLIT));

    int64_t r0 = 0, P0 = 0;
    int n_owned = 0, has_trail = 0, len = 0;
    if (active) {
        const int64_t total = nnz + rows;
        const int64_t s = w * ch;
        const int64_t e = (s + ch < total) ? s + ch : total;
        const int64_t ra = chunk_row[w];      // row holding item s
        const int64_t rb = chunk_row[w + 1];  // row holding item e (== rows after the last item)
        // first row: a short row that began in the previous chunk was finished there; a long one is continued from item s
        {
            const int64_t pa = ptr[ra], pa1 = ptr[ra + 1];
            const bool before = (pa + ra) < s;
            const bool is_long = (pa1 - pa + 1) > SPMM_SPLIT;
            if (before && !is_long) {
                r0 = ra + 1;
                P0 = pa1;
            } else {
                r0 = ra;
                P0 = before ? s - ra : pa;
            }
        }
        // last row: the row holding item e, if it starts inside this chunk, is finished here when short and cut (-> carry) when long
        int64_t r_stop, P1;
        if (rb < rows && (ptr[rb] + rb) < e) {
            const int64_t pb = ptr[rb], pb1 = ptr[rb + 1];
            r_stop = rb + 1;
            if ((pb1 - pb + 1) > SPMM_SPLIT) {
                P1 = (e - rb < pb1) ? e - rb : pb1;
                has_trail = 1;
            } else {
                P1 = pb1;
            }
        } else {
            r_stop = rb;
            P1 = (rb < rows) ? ptr[rb] : nnz;
        }
        if (r_stop < r0) r_stop = r0;
        const int nproc = (int)(r_stop - r0);
        n_owned = nproc - has_trail;
        for (int k = lane; k < nproc; k += WAVE) {
            int64_t en = ptr[r0 + k + 1];
            if (k == nproc - 1) en = P1;  // a cut row ends at the chunk end
            s_end[k] = (int32_t)(en - P0);
        }
        len = (int)(P1 - P0);
        if (len < 0) len = 0;
        for (int k = lane; k < len; k += WAVE) {
            const T a = val[P0 + k];
            SpEntry<T> en;
            en.c = col[P0 + k];
            en.v = conj_a ? vt<T>::conj(a) : a;
            s_nz[k] = en;
        }
    }
    __syncthreads();
    if (!active) return;

    const int g = lane / LPN;
    const int li = lane % LPN;
    const int nproc = n_owned + has_trail;
    __amdgpu_buffer_rsrc_t b_rsrc;
    if constexpr (TAG == SPMM_TAG_BUFFER) b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0xffffffff, 0x00020000);
    constexpr int E_NONE = 0x7fffffff;

    for (int64_t j0 = jlo; j0 < jhi; j0 += (int64_t)LPN * V) {
        const int64_t jc = j0 + (int64_t)li * V;
        const bool col_ok = jc < jhi;  // V divides N on the vector path
        const T* bcol = B + (col_ok ? jc : jlo) * b_cs;
        int k = 0;  // current row (wave-uniform); Ek = its end among the staged nonzeros
        int Ek = __builtin_amdgcn_readfirstlane(nproc > 0 ? s_end[0] : E_NONE);
        T acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = vt<T>::zero();
        auto fin = [&]() {  // row k ends here: its groups' sums are combined; C for an owned row, the carry for the cut one
#pragma unroll
            for (int off = LPN; off < WAVE; off <<= 1) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = vt<T>::add(acc[v], shfl_xor_val(acc[v], off));
            }
            if (g == 0 && col_ok) {
                if (k < n_owned) {
                    T* crow = C + (r0 + k) * c_rs + jc * c_cs;
                    vec<T, V> out;
                    if (beta_zero) {
#pragma unroll
                        for (int v = 0; v < V; ++v) out.v[v] = vt<T>::mul(alpha, acc[v]);
                    } else {
                        vec<T, V> old;
                        if (V > 1) old = *reinterpret_cast<const vec<T, V>*>(crow);
                        else old.v[0] = crow[0];
#pragma unroll
                        for (int v = 0; v < V; ++v) out.v[v] = vt<T>::fma(alpha, acc[v], vt<T>::mul(beta, old.v[v]));
                    }
                    // the output row is written once and not read again by this product: non-temporal stores keep it from
                    // displacing rows of B in the L2 (headline 1.47 -> 1.35 ms, row-owned form 1.84 -> 1.75 ms,
                    // profiles/r05_spmm_nt_store_ab.log; non-temporal loads of A changed nothing)
                    if constexpr (V * sizeof(T) == 16) nt_store16(crow, out.v);
                    else if (V > 1) *reinterpret_cast<vec<T, V>*>(crow) = out;
                    else crow[0] = out.v[0];
                } else {
                    T* cv = carry_val + w * N + jc;
                    if (V > 1) {
                        vec<T, V> out;
#pragma unroll
                        for (int v = 0; v < V; ++v) out.v[v] = acc[v];
                        *reinterpret_cast<vec<T, V>*>(cv) = out;
                    } else {
                        cv[0] = acc[0];
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) acc[v] = vt<T>::zero();
            ++k;
            Ek = __builtin_amdgcn_readfirstlane(k < nproc ? s_end[k] : E_NONE);
        };
        for (int p = 0; p < len; p += NG * U) {
            SpEntry<T> nz[U];
            vec<T, V> b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p + u * NG + g;
                nz[u] = s_nz[pp < len ? pp : len - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (TAG == SPMM_TAG_BUFFER) {
                    const int32_t cidx = nz[u].c & 0x7fffffff;
                    const unsigned voff = (unsigned)(((int64_t)cidx * b_rs + (col_ok ? jc : jlo)) * (int64_t)sizeof(T));
                    b[u] = spmm_tagged_load<T, V>(b_rsrc, voff, nz[u].c < 0);
                } else {
                    const T* src = bcol + (int64_t)nz[u].c * b_rs;
                    if (V > 1) b[u] = *reinterpret_cast<const vec<T, V>*>(src);
                    else b[u].v[0] = src[0];
                }
            }
            if (p + NG * U <= Ek) {  // no row ends inside this batch (Ek <= len): straight multiply-adds
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[v] = vt<T>::fma(nz[u].v, b[u].v[v], acc[v]);
                }
            } else {
                // row ends inside the batch: NG nonzeros at a time, the batch rotating through slot 0 (one copy of the
                // row-end code)
#pragma unroll 1
                for (int u = 0; u < U; ++u) {
                    const int q = p + u * NG;
                    if (q + NG <= Ek) {
#pragma unroll
                        for (int v = 0; v < V; ++v) acc[v] = vt<T>::fma(nz[0].v, b[0].v[v], acc[v]);
                    } else {
#pragma unroll 1
                        for (int gg = 0; gg < NG; ++gg) {
                            if (q + gg >= len) break;
                            while (q + gg >= Ek) fin();
                            if (g == gg) {
#pragma unroll
                                for (int v = 0; v < V; ++v) acc[v] = vt<T>::fma(nz[0].v, b[0].v[v], acc[v]);
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t + 1 < U; ++t) {
                        nz[t] = nz[t + 1];
                        b[t] = b[t + 1];
                    }
                }
            }
        }
        while (Ek <= len) fin();  // the rows that end with the chunk's last nonzero, empty rows included
    }
}

// ------------------------------------------------------------------------------------------------
// SpMV (N = 1): y := alpha * A x + beta * y  -- the lanes span NONZEROS instead of dense columns.
// Same work partition, ownership rules and carry / fix-up machinery as k_spmm.
// ------------------------------------------------------------------------------------------------
constexpr int SPMV_U = 4;  // nonzeros per lane in flight (k_spmv)

template <typename T>
__global__ void __launch_bounds__(SPMM_WAVES* WAVE)
    k_spmv(int64_t rows, int64_t nnz, const int64_t* __restrict__ ptr, const int32_t* __restrict__ col,
           const T* __restrict__ val, const SpmmChunk* __restrict__ chunk_desc, int64_t nchunks, int ch, int conj_a,
           const T* __restrict__ x, int64_t x_s, T* __restrict__ y, int64_t y_s, T alpha, T beta, int beta_zero,
           T* __restrict__ carry_val)
{
    MI_DYN_SMEM(smem);
    const int wave_in_block = threadIdx.x / WAVE;
    const int lane = threadIdx.x % WAVE;
    const int64_t w = (int64_t)blockIdx.x * SPMM_WAVES + wave_in_block;
    const bool active = w < nchunks;
    const size_t per_wave = ((size_t)(ch + SPMM_SPLIT) * sizeof(T) + (size_t)(ch + 2) * sizeof(int32_t) + 15) & ~size_t(15);
    char* base = smem + per_wave * wave_in_block;
    T* s_prod = reinterpret_cast<T*>(base);
    int32_t* s_end = reinterpret_cast<int32_t*>(base + sizeof(T) * (size_t)(ch + SPMM_SPLIT));

    int64_t r0 = 0;
    int n_owned = 0, has_trail = 0, nproc = 0;
    if (active) {
        // identical partition to k_spmm: the chunk's descriptor (k_spmm_plan_desc)
        const SpmmChunk d = chunk_desc[w];
        r0 = d.r0;
        const int64_t P0 = d.P0, P1 = d.P1;
        has_trail = d.has_trail;
        nproc = d.nproc;
        n_owned = nproc - has_trail;
        for (int k = lane; k < nproc; k += WAVE) {
            int64_t en = ptr[r0 + k + 1];
            if (k == nproc - 1) en = P1;
            s_end[k] = (int32_t)(en - P0);
        }
        const int len = (int)(P1 - P0);
        // coalesced A stream + gather of x, four nonzeros per lane in flight (all loads of a group are issued
        // before the first use: the chain col -> x is paid once per group, not once per nonzero)
        for (int k0 = lane; k0 < len; k0 += SPMV_U * WAVE) {
            // loads unconditional (positions past the end re-read the lane's first item), every use after the last load
            T a[SPMV_U], xv[SPMV_U];
            int32_t cc[SPMV_U];
#pragma unroll
            for (int u = 0; u < SPMV_U; ++u) {
                const int k = k0 + u * WAVE < len ? k0 + u * WAVE : k0;
                cc[u] = col[P0 + k];
                a[u] = val[P0 + k];
            }
#pragma unroll
            for (int u = 0; u < SPMV_U; ++u) xv[u] = x[(int64_t)cc[u] * x_s];  // plain loads: non-temporal ones are 1.7-2x slower here (measured)
#pragma unroll
            for (int u = 0; u < SPMV_U; ++u) {
                const int k = k0 + u * WAVE;
                if (k < len) s_prod[k] = vt<T>::mul(conj_a ? vt<T>::conj(a[u]) : a[u], xv[u]);
            }
        }
    }
    __syncthreads();
    if (!active) return;

    // pass 1: one lane per row for segments of at most 32 products
    for (int k = lane; k < nproc; k += WAVE) {
        const int begin = k ? s_end[k - 1] : 0;
        const int end = s_end[k];
        if (end - begin <= 32) {
            T sum = vt<T>::zero();
            for (int p = begin; p < end; ++p) sum = vt<T>::add(sum, s_prod[p]);
            if (k < n_owned) {
                T* yy = y + (r0 + k) * y_s;
                *yy = beta_zero ? vt<T>::mul(alpha, sum) : vt<T>::fma(alpha, sum, vt<T>::mul(beta, *yy));
            } else {
                carry_val[w] = sum;
            }
        }
    }
    // pass 2: longer segments, the whole wave per segment
    for (int k = 0; k < nproc; ++k) {
        const int begin = k ? s_end[k - 1] : 0;
        const int end = s_end[k];
        if (end - begin <= 32) continue;  // wave-uniform
        T sum = vt<T>::zero();
        for (int p = begin + lane; p < end; p += WAVE) sum = vt<T>::add(sum, s_prod[p]);
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) sum = vt<T>::add(sum, shfl_xor_val(sum, d));
        if (lane == 0) {
            if (k < n_owned) {
                T* yy = y + (r0 + k) * y_s;
                *yy = beta_zero ? vt<T>::mul(alpha, sum) : vt<T>::fma(alpha, sum, vt<T>::mul(beta, *yy));
            } else {
                carry_val[w] = sum;
            }
        }
    }
}

// add the carries of every cut row to the row its owner wrote.  The schedule (which chunks carry into which row) depends
// only on A and the chunk size, so it is precomputed in the plan (k_spmm_plan_tasks: a list of short tasks, a list of
// long ones).  Workgroups of eight lane groups (LPN lanes x V values); the first ceil(short / 8) workgroups of the
// (virtual) grid take eight short tasks each, the others one long task each (round 5; one task of either kind per
// workgroup before -- the column-partitioned product cuts ~50 000 sub-rows of two or three carries each, and a workgroup
// with two barriers per task spent 68 us on them):
//   * a short task belongs to ONE lane group: carries in chunk order, four loads in flight, one read-modify-write of C,
//     no LDS, no barrier;
//   * a long task (a hub row is cut into hundreds of chunks) is shared by the eight groups: group g takes the chunks
//     first + g, first + g + 8, ..., group 0 combines the eight partial sums through LDS in a FIXED tree.
// Which form a task takes depends on its length only: the same bits on every call.
template <typename T, int V, int LPN>
__global__ void __launch_bounds__(8 * LPN)
    k_spmm_fixup(const unsigned long long* __restrict__ n_tasks_dev, const int32_t* __restrict__ tasks, int64_t cap,
                 const T* __restrict__ carry_val, int64_t N, T* __restrict__ C, int64_t c_rs, int64_t c_cs, T alpha)
{
    constexpr int G = 8, U = 4;
    __shared__ T part_s[G][LPN * V];
    const int g = threadIdx.x / LPN, li = threadIdx.x % LPN;
    // the counts live on the device (the grid may be an upper bound, and is capped)
    const int64_t n_short = (int64_t)n_tasks_dev[0], n_long = (int64_t)n_tasks_dev[1];
    const int64_t short_blocks = (n_short + G - 1) / G;
    for (int64_t blk = blockIdx.x; blk < short_blocks + n_long; blk += gridDim.x) {
        if (blk < short_blocks) {
            const int64_t task = blk * G + g;
            if (task >= n_short) continue;
            const int64_t row = tasks[3 * task], first = tasks[3 * task + 1], last = tasks[3 * task + 2];
            for (int64_t j0 = 0; j0 < N; j0 += (int64_t)LPN * V) {
                const int64_t jc = j0 + (int64_t)li * V;
                if (jc >= N) continue;  // V divides N on the vector path
                T sum[V];
#pragma unroll
                for (int v = 0; v < V; ++v) sum[v] = vt<T>::zero();
                for (int64_t u = first; u <= last; u += U) {
                    vec<T, V> cvv[U];
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        const int64_t uu = (u + k <= last) ? u + k : last;
                        if (V > 1) cvv[k] = *reinterpret_cast<const vec<T, V>*>(carry_val + uu * N + jc);
                        else cvv[k].v[0] = carry_val[uu * N + jc];
                    }
#pragma unroll
                    for (int k = 0; k < U; ++k) {
                        if (u + k <= last) {
#pragma unroll
                            for (int v = 0; v < V; ++v) sum[v] = vt<T>::add(sum[v], cvv[k].v[v]);
                        }
                    }
                }
                T* c = C + row * c_rs + jc * c_cs;
                if (V > 1) {
                    vec<T, V> old = *reinterpret_cast<const vec<T, V>*>(c);
#pragma unroll
                    for (int v = 0; v < V; ++v) old.v[v] = vt<T>::fma(alpha, sum[v], old.v[v]);
                    *reinterpret_cast<vec<T, V>*>(c) = old;
                } else {
                    *c = vt<T>::fma(alpha, sum[0], *c);
                }
            }
            continue;
        }
        const int64_t task = cap - 1 - (blk - short_blocks);  // the long tasks fill the list from its back
        const int64_t row = tasks[3 * task], first = tasks[3 * task + 1], last = tasks[3 * task + 2];
        for (int64_t j0 = 0; j0 < N; j0 += (int64_t)LPN * V) {
            const int64_t jc = j0 + (int64_t)li * V;
            const bool col_ok = jc < N;  // V divides N on the vector path
            const int64_t jl = col_ok ? jc : j0;
            T part[V];
#pragma unroll
            for (int v = 0; v < V; ++v) part[v] = vt<T>::zero();
            for (int64_t u = first + g; u <= last; u += (int64_t)G * U) {
                vec<T, V> cvv[U];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int64_t uu = (u + (int64_t)k * G <= last) ? u + (int64_t)k * G : last;  // clamp; masked out below
                    if (V > 1) cvv[k] = *reinterpret_cast<const vec<T, V>*>(carry_val + uu * N + jl);
                    else cvv[k].v[0] = carry_val[uu * N + jl];
                }
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    if (u + (int64_t)k * G <= last) {
#pragma unroll
                        for (int v = 0; v < V; ++v) part[v] = vt<T>::add(part[v], cvv[k].v[v]);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < V; ++v) part_s[g][li * V + v] = part[v];
            __syncthreads();
            if (g == 0 && col_ok) {
                T sum[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int x = li * V + v;
                    sum[v] = vt<T>::add(vt<T>::add(vt<T>::add(part_s[0][x], part_s[1][x]), vt<T>::add(part_s[2][x], part_s[3][x])),
                                        vt<T>::add(vt<T>::add(part_s[4][x], part_s[5][x]), vt<T>::add(part_s[6][x], part_s[7][x])));
                }
                T* c = C + row * c_rs + jc * c_cs;
                if (V > 1) {
                    vec<T, V> old = *reinterpret_cast<const vec<T, V>*>(c);
#pragma unroll
                    for (int v = 0; v < V; ++v) old.v[v] = vt<T>::fma(alpha, sum[v], old.v[v]);
                    *reinterpret_cast<vec<T, V>*>(c) = old;
                } else {
                    *c = vt<T>::fma(alpha, sum[0], *c);
                }
            }
            __syncthreads();
        }
    }
}

// dense layout conversion through a 32 x 32 LDS tile: element (i, j) moves from
// src[i * s_rs + j * s_cs] to dst[i * d_rs + j * d_cs]; reads and writes are both coalesced along
// whichever index is contiguous on that side.  Used to run column-major operands through the
// row-major (fully coalesced) SpMM kernel.
template <typename T>
__global__ void __launch_bounds__(256)
    k_convert_layout(int64_t rows, int64_t cols, const T* __restrict__ src, int64_t s_rs, int64_t s_cs,
                     T* __restrict__ dst, int64_t d_rs, int64_t d_cs)
{
    __shared__ T tile[32][33];  // [j local][i local]
    const int tx = threadIdx.x % 32, ty = threadIdx.x / 32;  // 32 x 8
    const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;  // the long (row) dimension on grid.x: grid.y stops at 65535
    for (int k = 0; k < 4; ++k) {
        const int a = tx, b = ty + 8 * k;
        const int il = (s_rs == 1) ? a : b, jl = (s_rs == 1) ? b : a;
        const int64_t i = i0 + il, j = j0 + jl;
        if (i < rows && j < cols) tile[jl][il] = src[i * s_rs + j * s_cs];
    }
    __syncthreads();
    for (int k = 0; k < 4; ++k) {
        const int a = tx, b = ty + 8 * k;
        const int jl = (d_cs == 1) ? a : b, il = (d_cs == 1) ? b : a;
        const int64_t i = i0 + il, j = j0 + jl;
        if (i < rows && j < cols) dst[i * d_rs + j * d_cs] = tile[jl][il];
    }
}

template <typename T>
static void convert_layout(int64_t rows, int64_t cols, const T* src, int64_t s_rs, int64_t s_cs, T* dst, int64_t d_rs,
                           int64_t d_cs)
{
    if (rows == 0 || cols == 0) return;
    MI_LAUNCH((k_convert_layout<T>), dim3((unsigned)ceil_div(rows, 32), (unsigned)ceil_div(cols, 32)), dim3(256),
              ctx().stream, rows, cols, src, s_rs, s_cs, dst, d_rs, d_cs);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// ---- hot / cold analysis, entirely on the device ---------------------------------------------------
// (1) column histogram over a SAMPLE of the nonzeros (every `stride`-th block of 2048 consecutive ones:
//     the hot set only needs the ranking of the heavy columns, and an exact count costs one L2 atomic per
//     nonzero -- 3 ms at 31 M nonzeros, more than the product it serves);
// (2) histogram of the (clamped) counts -> the count threshold whose column set fits the budget, and the
//     share of sampled nonzeros it covers;  (3) tagged copy of the column indices.
// Nothing is read back synchronously: the decision word travels to the host with an asynchronous copy
// that a later call polls (AsyncWord).
constexpr int HOT_BLOCK = 2048;   // nonzeros per sampled block
constexpr int HOT_BINS = 4096;    // count histogram bins (counts >= HOT_BINS - 1 share the last bin)

__global__ void __launch_bounds__(256)
    k_count_cols(const int32_t* __restrict__ col, int64_t nnz, int64_t stride, unsigned* __restrict__ counts)
{
    const int64_t base = (int64_t)blockIdx.x * stride * HOT_BLOCK;
    const int64_t end = base + HOT_BLOCK < nnz ? base + HOT_BLOCK : nnz;
    for (int64_t i = base + threadIdx.x; i < end; i += 256) atomicAdd(&counts[col[i]], 1u);
}

// hist[b] = columns with min(count, HOT_BINS - 1) == b;  wsum[b] = sum of their counts
__global__ void __launch_bounds__(256)
    k_count_hist(const unsigned* __restrict__ counts, int64_t ncols, unsigned long long* __restrict__ hist,
                 unsigned long long* __restrict__ wsum)
{
    __shared__ unsigned h[HOT_BINS];
    __shared__ unsigned long long big;  // exact sum of the counts that land in the last bin
    for (int k = threadIdx.x; k < HOT_BINS; k += 256) h[k] = 0u;
    if (threadIdx.x == 0) big = 0ull;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncols; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned c = counts[i];
        if (c == 0) continue;
        if (c >= HOT_BINS - 1) {
            atomicAdd(&h[HOT_BINS - 1], 1u);
            atomicAdd(&big, (unsigned long long)c);
        } else {
            atomicAdd(&h[c], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < HOT_BINS; k += 256) {
        if (h[k]) {
            atomicAdd(&hist[k], (unsigned long long)h[k]);
            if (k < HOT_BINS - 1) atomicAdd(&wsum[k], (unsigned long long)h[k] * (unsigned long long)k);
        }
    }
    if (threadIdx.x == 0 && big) atomicAdd(&wsum[HOT_BINS - 1], big);
}

// one workgroup: decision = {flag, threshold, hot columns, covered (sampled) nonzeros, sampled nonzeros}
__global__ void __launch_bounds__(1024)
    k_pick_threshold(const unsigned long long* __restrict__ hist, const unsigned long long* __restrict__ wsum,
                     int64_t hot_rows, int force, int64_t* __restrict__ decision)
{
    // suffix sums over the bins (4 bins per thread, then a block scan of the thread totals): N(k) = columns with
    // count >= k, W(k) = nonzeros on them.  The threshold is the smallest k >= 1 whose set still fits the budget.
    __shared__ unsigned long long cnt[HOT_BINS], wt[HOT_BINS];
    __shared__ unsigned long long tn[1024], tw[1024];
    __shared__ int best;
    const int tid = threadIdx.x;
    for (int k = tid; k < HOT_BINS; k += 1024) {
        cnt[k] = k ? hist[k] : 0ull;
        wt[k] = k ? wsum[k] : 0ull;
    }
    if (tid == 0) best = HOT_BINS;
    __syncthreads();
    constexpr int PER = HOT_BINS / 1024;
    // thread t owns bins [HOT_BINS - PER (t + 1), HOT_BINS - PER t): thread 0 the highest counts
    unsigned long long ln = 0, lw = 0;
    for (int u = 0; u < PER; ++u) {
        const int k = HOT_BINS - 1 - (tid * PER + u);
        ln += cnt[k];
        lw += wt[k];
    }
    tn[tid] = ln;
    tw[tid] = lw;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan: totals of the threads holding higher counts
        unsigned long long a = 0, b = 0;
        if (tid >= off) {
            a = tn[tid - off];
            b = tw[tid - off];
        }
        __syncthreads();
        tn[tid] += a;
        tw[tid] += b;
        __syncthreads();
    }
    const unsigned long long total = tw[1023];
    unsigned long long n_run = tn[tid] - ln, w_run = tw[tid] - lw;  // everything above this thread's bins
    for (int u = 0; u < PER; ++u) {
        const int k = HOT_BINS - 1 - (tid * PER + u);
        n_run += cnt[k];
        w_run += wt[k];
        cnt[k] = n_run;  // now N(k)
        wt[k] = w_run;   // now W(k)
    }
    __syncthreads();
    // N is non-increasing in k: candidates are the k with a non-empty bin whose set fits; take the smallest.  A single
    // over-full top class (ties) is accepted here and judged by the "2 x budget" rule below.
    for (int u = 0; u < PER; ++u) {
        const int k = HOT_BINS - 1 - (tid * PER + u);
        if (k < 1) continue;
        const unsigned long long above = (k + 1 < HOT_BINS) ? cnt[k + 1] : 0ull;
        const bool nonempty = cnt[k] != above;
        if (nonempty && ((int64_t)cnt[k] <= hot_rows || above == 0)) atomicMin(&best, k);
    }
    __syncthreads();
    if (tid != 0) return;
    const int thr = best;
    const unsigned long long n = thr < HOT_BINS ? cnt[thr] : 0ull, cov = thr < HOT_BINS ? wt[thr] : 0ull;
    const double share = total ? (double)cov / (double)total : 0.0;
    // worth it only when a small set takes a real share of the gather and ties did not blow the set up
    int flag = (thr < HOT_BINS && thr >= 2 && share >= 0.10 && (int64_t)n <= 2 * hot_rows) ? 1 : 0;
    if (force && thr < HOT_BINS) flag = 1;
    decision[0] = flag;
    decision[1] = thr;
    decision[2] = (int64_t)n;
    decision[3] = (int64_t)cov;
    decision[4] = (int64_t)total;
}

__global__ void k_tag_cols(const int32_t* __restrict__ col, int64_t nnz, const unsigned* __restrict__ counts,
                           const int64_t* __restrict__ decision, int32_t* __restrict__ tagged)
{
    if (decision[0] == 0) return;  // no useful hot set: the tagged copy is never used
    const unsigned threshold = (unsigned)decision[1];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = col[i];
        tagged[i] = counts[c] >= threshold ? c : (int32_t)((unsigned)c | 0x80000000u);
    }
}

// Enqueue the analysis for `hot_rows` rows of B behind whatever is already on the stream.
static void enqueue_hot_analysis(SpmmPlan& p, const Csr& m, int64_t hot_rows)
{
    Context& c = ctx();
    p.hot_rows_budget = hot_rows;
    p.tagged = false;
    p.hot_coverage = 0.0;
    p.hot_state = 2;  // "decided: untagged" unless the analysis below is launched
    const bool force = options().spmm_hot_force != 0;
    if (hot_rows <= 0 || m.nnz == 0) return;
    if (hot_rows > m.cols) hot_rows = m.cols;
    if (!force && (m.nnz < (int64_t)1 << 20 || m.cols <= 4 * hot_rows)) return;  // small / everything fits
    const int64_t nblocks = ceil_div(m.nnz, HOT_BLOCK);
    const int64_t stride = (m.nnz >= (int64_t)1 << 23) ? 8 : 1;  // sample 1/8 of the big ones
    unsigned* counts = static_cast<unsigned*>(c.scratch_alloc(sizeof(unsigned) * (size_t)m.cols));
    unsigned long long* hist = static_cast<unsigned long long*>(c.scratch_alloc(sizeof(unsigned long long) * 2 * HOT_BINS));
    MI_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(unsigned) * (size_t)m.cols, c.stream));
    MI_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(unsigned long long) * 2 * HOT_BINS, c.stream));
    MI_LAUNCH(k_count_cols, dim3((unsigned)ceil_div(nblocks, stride)), dim3(256), c.stream, (const int32_t*)m.col, m.nnz,
              stride, counts);
    const int64_t hgrid = ceil_div(m.cols, 256) < 1024 ? ceil_div(m.cols, 256) : 1024;
    MI_LAUNCH(k_count_hist, dim3((unsigned)hgrid), dim3(256), c.stream, (const unsigned*)counts, m.cols, hist,
              hist + HOT_BINS);
    if (!p.hot_decision.p) p.hot_decision.alloc(sizeof(int64_t) * 8);
    MI_LAUNCH(k_pick_threshold, dim3(1), dim3(1024), c.stream, (const unsigned long long*)hist,
              (const unsigned long long*)(hist + HOT_BINS), hot_rows, (int)force, p.hot_decision.as<int64_t>());
    p.col_tagged.alloc(sizeof(int32_t) * (size_t)m.nnz);
    const int64_t tg = ceil_div(m.nnz, 256) < (1 << 16) ? ceil_div(m.nnz, 256) : (1 << 16);
    MI_LAUNCH(k_tag_cols, dim3((unsigned)tg), dim3(256), c.stream, (const int32_t*)m.col, m.nnz, (const unsigned*)counts,
              (const int64_t*)p.hot_decision.as<int64_t>(), p.col_tagged.as<int32_t>());
    p.hot_word.post(p.hot_decision.p, sizeof(int64_t) * 5, c.stream);
    p.hot_state = 1;
}

// adopt a finished analysis (non-blocking unless `wait`)
static void poll_hot_analysis(SpmmPlan& p, bool wait)
{
    if (p.hot_state != 1) return;
    if (wait) ctx().sync();
    if (!p.hot_word.ready()) return;
    const int64_t* d = p.hot_word.host;
    p.tagged = d[0] != 0;
    p.hot_coverage = d[4] ? (double)d[3] / (double)d[4] : 0.0;
    if (!p.tagged) p.col_tagged.release();
    p.hot_state = 2;
}

static SpmmPlan& get_plan(SpmmPlan& p, std::mutex& mtx, const Csr& m, int chunk, int64_t hot_rows)
{
    std::lock_guard<std::mutex> lk(mtx);
    Context& c = ctx();
    if (!(p.chunk == chunk && p.chunk_row.p)) {
        // the row partition and the fix-up schedule: two small kernels, nothing read back here
        const int64_t total = m.nnz + m.rows;
        p.nchunks = total > 0 ? ceil_div(total, chunk) : 1;
        p.chunk_row.alloc(sizeof(int32_t) * (size_t)(p.nchunks + 1));
        MI_LAUNCH(k_spmm_plan, dim3((unsigned)ceil_div(p.nchunks + 1, 256)), dim3(256), c.stream,
                  (const int64_t*)m.ptr, m.rows, m.nnz, (int64_t)chunk, p.nchunks, p.chunk_row.as<int32_t>());
        p.chunk = chunk;
        p.tasks.alloc(sizeof(int32_t) * 3 * (size_t)(p.nchunks + 1));
        p.n_tasks_dev.alloc(2 * sizeof(unsigned long long));
        MI_HIP_CHECK(hipMemsetAsync(p.n_tasks_dev.p, 0, 2 * sizeof(unsigned long long), c.stream));
        MI_LAUNCH(k_spmm_plan_tasks, dim3((unsigned)ceil_div(p.nchunks, 256)), dim3(256), c.stream, (const int64_t*)m.ptr,
                  (const int32_t*)p.chunk_row.as<int32_t>(), m.rows, m.nnz, (int64_t)chunk, p.nchunks,
                  p.tasks.as<int32_t>(), static_cast<unsigned long long*>(p.n_tasks_dev.p));
        p.chunk_desc.alloc(sizeof(SpmmChunk) * (size_t)(p.nchunks + 1));
        MI_LAUNCH(k_spmm_plan_desc, dim3((unsigned)ceil_div(p.nchunks, 256)), dim3(256), c.stream, (const int64_t*)m.ptr,
                  (const int32_t*)p.chunk_row.as<int32_t>(), m.rows, m.nnz, (int64_t)chunk, p.nchunks,
                  p.chunk_desc.as<SpmmChunk>());
        p.n_tasks = p.n_tasks_long = -1;
        p.n_tasks_word.post(p.n_tasks_dev.p, 2 * sizeof(unsigned long long), c.stream);
        p.uses = 0;
        counters().spmm_plans_built += 1.0;
    }
    if (p.n_tasks < 0 && p.n_tasks_word.ready()) {
        p.n_tasks = p.n_tasks_word.host[0];
        p.n_tasks_long = p.n_tasks_word.host[1];
    }
    // hot / cold tags: analysed behind the SECOND product of a handle (a single-use handle never pays),
    // or at once when a test / tool asks for the synchronous form
    const bool sync_plan = options().spmm_plan_sync != 0 || options().spmm_hot_force != 0;
    if (p.hot_rows_budget != hot_rows && (p.hot_rows_budget >= 0 || sync_plan)) {
        p.reset_hot();  // the budget changed (another row width): analyse again, now
        enqueue_hot_analysis(p, m, hot_rows);
    }
    poll_hot_analysis(p, sync_plan);
    return p;
}

// called after a product has been enqueued: count the use and start the analysis when reuse is proven
// (`hold_hot`: the column-partitioned form may still replace this plan -- the analysis waits until that is decided)
static void plan_after_product(SpmmPlan& p, std::mutex& mtx, const Csr& m, int64_t hot_rows, bool hold_hot)
{
    std::lock_guard<std::mutex> lk(mtx);
    ++p.uses;
    if (p.uses >= 2 && !hold_hot && p.hot_state == 0 && p.hot_rows_budget < 0) enqueue_hot_analysis(p, m, hot_rows);
}

template <typename T, int V, int LPN, int U>
static void launch_spmm_u(const Csr& m, const SpmmPlan& p, int conj_a, const T* B, int64_t b_rs, int64_t b_cs, T* C,
                          int64_t c_rs, int64_t c_cs, int64_t N, T alpha, T beta, T* carry_val,
                          int tag_mode, int slices, const SpmmParts& parts)
{
    Context& c = ctx();
    const size_t per_wave = (spmm_wave_lds<T>(p.chunk) + 15) & ~size_t(15);
    const size_t lds = per_wave * SPMM_WAVES;
    unsigned grid = (unsigned)ceil_div(p.nchunks, SPMM_WAVES);
    if (parts.P > 0) {  // eight interleaved block lists, each as long as the longest partition's
        int64_t mx = 0;
        for (int q = 0; q < parts.P; ++q) mx = std::max(mx, parts.cs[q + 1] - parts.cs[q]);
        grid = (unsigned)ceil_div(mx, SPMM_WAVES) * 8u;
        if (grid == 0) return;
    } else if (slices > 1) grid = (unsigned)ceil_div((int64_t)grid, 8 / slices) * 8u;  // see the block mapping in k_spmm
    const int beta_zero = vt<T>::is_zero(beta) ? 1 : 0;
    note_kernel("mi::k_spmm<%s, V=%d, LPN=%d, U=%d, TAG=%d> x %d column slice%s%s", type_name<T>(), V, LPN, U,
                (V * sizeof(T) == 16) ? tag_mode : 0, slices, slices > 1 ? "s" : "",
                parts.P > 0 ? " (long rows by column partition + short rows row-owned)" : "");
#define MI_SPMM_LAUNCH(TAGMODE, COLS)                                                                                  \
    MI_LAUNCH_SMEM((k_spmm<T, V, LPN, U, TAGMODE>), dim3(grid), dim3(SPMM_WAVES * WAVE), lds, c.stream, m.rows, m.nnz, \
                   (const int64_t*)m.ptr, (const int32_t*)(COLS), (const T*)m.val,                                      \
                   (const int32_t*)p.chunk_row.as<int32_t>(), p.nchunks, p.chunk, conj_a, B, b_rs, b_cs, C, c_rs,      \
                   c_cs, N, alpha, beta, beta_zero, carry_val, slices, parts)
    if constexpr (V * sizeof(T) == 16) {
        if (tag_mode == SPMM_TAG_BUFFER) {
            MI_SPMM_LAUNCH(SPMM_TAG_BUFFER, p.col_tagged.as<int32_t>());
            return;
        }
    }
    MI_SPMM_LAUNCH(SPMM_TAG_NONE, m.col);
#undef MI_SPMM_LAUNCH
}

template <typename T, int V, int LPN>
static void launch_spmm(const Csr& m, const SpmmPlan& p, int conj_a, const T* B, int64_t b_rs, int64_t b_cs, T* C,
                        int64_t c_rs, int64_t c_cs, int64_t N, T alpha, T beta, T* carry_val,
                        int tag_mode, int slices, const SpmmParts& parts)
{
    // U = independent 16-byte loads in flight per lane.  The deeper variant exists for the 512-byte
    // row shapes of the headline configs only (keeps the instantiation count down).
    if constexpr (V > 1 && LPN >= 32) {
        if (options().spmm_unroll == 8) {
            launch_spmm_u<T, V, LPN, 8>(m, p, conj_a, B, b_rs, b_cs, C, c_rs, c_cs, N, alpha, beta, carry_val,
                                        tag_mode, slices, parts);
            return;
        }
    }
    launch_spmm_u<T, V, LPN, 4>(m, p, conj_a, B, b_rs, b_cs, C, c_rs, c_cs, N, alpha, beta, carry_val, tag_mode,
                                slices, parts);
}

// every byte offset (row * ld + column) * elem of the operand, plus the 16 bytes one load covers, fits 32 bits
static inline bool dense_bytes_below_4g(int64_t rows, int64_t ld, size_t elem)
{
    return (double)rows * (double)ld * (double)elem + 16.0 < 4294967295.0;
}

// ------------------------------------------------------------------------------------------------
// column-partitioned plan (SpmmKpart, common.hpp): build
// ------------------------------------------------------------------------------------------------
// part(column): a multiplicative hash, so that every partition gets the same share of the heavy columns of a power-law
// matrix whatever their numbering (measured on the headline R-MAT: partitions within 1.5 % of each other, the same as a
// round-robin over the columns ranked by count -- profiles/r05_spmm_kpart_probe.log)
__host__ __device__ __forceinline__ int kp_part(int32_t c, int P) { return (int)((((uint32_t)c * 0x9E3779B1u) >> 16) & (uint32_t)(P - 1)); }

__global__ void k_kp_flag(const int64_t* __restrict__ ptr, int64_t rows, int64_t min_row, int64_t* __restrict__ flag,
                          int64_t* __restrict__ short_len)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t len = ptr[r + 1] - ptr[r];
    const bool lg = len >= min_row;
    flag[r] = lg ? 1 : 0;
    short_len[r] = lg ? 0 : len;
}

__global__ void k_kp_rowid(const int64_t* __restrict__ flag, const int64_t* __restrict__ lidx, int64_t rows,
                           int32_t* __restrict__ rowid)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows && flag[r]) rowid[lidx[r]] = (int32_t)r;
}

// one wave per long row: entries per partition -> cnt[p * n_long + i]
__global__ void __launch_bounds__(256)
    k_kp_count(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const int32_t* __restrict__ rowid,
               int64_t n_long, int P, int64_t* __restrict__ cnt)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (i >= n_long) return;
    const int64_t r = rowid[i], b = ptr[r], e = ptr[r + 1];
    int mine[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t k = b + lane; k < e; k += WAVE) {
        const int q = kp_part(col[k], P);
#pragma unroll
        for (int t = 0; t < 8; ++t) mine[t] += (q == t) ? 1 : 0;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        int v = mine[t];
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0 && t < P) cnt[(int64_t)t * n_long + i] = v;
    }
}

// one wave per long row: stable split of its entries into the P sub-rows (the order inside a sub-row is the row's own)
template <typename W>
__global__ void __launch_bounds__(256)
    k_kp_fill_long(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const W* __restrict__ val,
                   const int32_t* __restrict__ rowid, int64_t n_long, int P, const int64_t* __restrict__ cat_ptr,
                   int32_t* __restrict__ cat_col, W* __restrict__ cat_val)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (i >= n_long) return;
    const int64_t r = rowid[i], b = ptr[r], e = ptr[r + 1];
    int64_t run[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) run[t] = t < P ? cat_ptr[(int64_t)t * n_long + i] : 0;
    for (int64_t k0 = b; k0 < e; k0 += WAVE) {
        const int64_t k = k0 + lane;
        const bool ok = k < e;
        const int32_t c = ok ? col[k] : 0;
        const int q = ok ? kp_part(c, P) : -1;
        int64_t dst = -1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            int total;
            const int rk = wave_rank(q == t, total);
            if (q == t) dst = run[t] + rk;
            run[t] += total;
        }
        if (ok) {
            cat_col[dst] = c;
            cat_val[dst] = val[k];
        }
    }
}

// eight lanes per short row: its entries move to the compacted arrays as they are
template <typename W>
__global__ void __launch_bounds__(256)
    k_kp_fill_short(const int64_t* __restrict__ ptr, const int32_t* __restrict__ col, const W* __restrict__ val,
                    int64_t rows, int64_t min_row, const int64_t* __restrict__ s_ptr, int32_t* __restrict__ s_col,
                    W* __restrict__ s_val)
{
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8;
    const int li = threadIdx.x % 8;
    if (r >= rows) return;
    const int64_t b = ptr[r], len = ptr[r + 1] - b;
    if (len >= min_row) return;
    const int64_t d = s_ptr[r];
    for (int64_t k = li; k < len; k += 8) {
        s_col[d + k] = col[b + k];
        s_val[d + k] = val[b + k];
    }
}

struct KpWord16 {
    uint64_t a, b;
};

// (the partial rows are read once, from HBM: non-temporal loads -- 59.8 -> 57.2 us on the headline matrix, product - 0.6 %;
// a non-temporal store of the result on top changes nothing: profiles/r06_kp_combine_nt_ab.log)
#ifndef MI_KP_COMBINE_NT
#define MI_KP_COMBINE_NT 1
#endif
// C[rowid[i]] += alpha * (partial[0 * n_long + i] + ... + partial[(P - 1) * n_long + i]): one lane group per long row, the P
// partial rows in flight together, summed in partition order -- the same bits on every call
template <typename T, int V, int LPN>
__global__ void __launch_bounds__(256)
    k_kp_combine(const T* __restrict__ partial, int64_t n_long, int P, const int32_t* __restrict__ rowid, int64_t N,
                 T* __restrict__ C, int64_t c_rs, T alpha, int overwrite)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPN;
    const int li = threadIdx.x % LPN;
    if (i >= n_long) return;
    T* crow = C + (int64_t)rowid[i] * c_rs;
    for (int64_t j = (int64_t)li * V; j < N; j += (int64_t)LPN * V) {  // V divides N on this path
        vec<T, V> part[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < P) {
#if MI_KP_COMBINE_NT
                if constexpr (sizeof(vec<T, V>) == 16) {
                    const u32x4 w = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(partial + ((int64_t)t * n_long + i) * N + j));
                    __builtin_memcpy(&part[t], &w, 16);
                } else
#endif
                part[t] = *reinterpret_cast<const vec<T, V>*>(partial + ((int64_t)t * n_long + i) * N + j);
            }
        // beta == 0 (`overwrite`): the row holds the zeros the short-row kernel wrote for it -- not read back, the result is
        // alpha * sum as the row-owned kernel forms it
        vec<T, V> out;
        if (!overwrite) out = *reinterpret_cast<const vec<T, V>*>(crow + j);
        T sum[V];
#pragma unroll
        for (int v = 0; v < V; ++v) sum[v] = part[0].v[v];
#pragma unroll
        for (int t = 1; t < 8; ++t)
            if (t < P) {
#pragma unroll
                for (int v = 0; v < V; ++v) sum[v] = vt<T>::add(sum[v], part[t].v[v]);
            }
#pragma unroll
        for (int v = 0; v < V; ++v) out.v[v] = overwrite ? vt<T>::mul(alpha, sum[v]) : vt<T>::fma(alpha, sum[v], out.v[v]);
#if MI_KP_COMBINE_NT >= 2
        if constexpr (sizeof(vec<T, V>) == 16) nt_store16(reinterpret_cast<T*>(crow + j), reinterpret_cast<const T*>(&out));
        else
#endif
        *reinterpret_cast<vec<T, V>*>(crow + j) = out;
    }
}

// Build the column-partitioned form of `m` (synchronous: three scans whose totals size the arrays; ~1 ms of device time at
// the headline's 31 M entries, paid once, ahead of the third product of a handle).  Sets p.kpart_state to 1 (declined) or 2.
static void build_kpart_impl(SpmmPlan& p, const Csr& m, char vtype)
{
    Context& c = ctx();
    const Options& o = options();
    const int P = (int)o.spmm_kpart_parts;
    const int64_t min_row = o.spmm_kpart_min_row;
    p.kpart_state = 1;
    p.kpart.reset();
    if (m.rows == 0 || m.nnz == 0) return;
    struct Events {  // destroyed on every way out (an allocation below may throw)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events()
        {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } evs;
    MI_HIP_CHECK(hipEventCreate(&evs.e0));
    MI_HIP_CHECK(hipEventCreate(&evs.e1));
    hipEvent_t e0 = evs.e0, e1 = evs.e1;
    MI_HIP_CHECK(hipEventRecord(e0, c.stream));
    const size_t vb = value_bytes(vtype);
    DevBuf flag_b, lidx_b, slen_b;
    flag_b.alloc(sizeof(int64_t) * (size_t)(m.rows + 1));
    lidx_b.alloc(sizeof(int64_t) * (size_t)(m.rows + 1));
    slen_b.alloc(sizeof(int64_t) * (size_t)(m.rows + 1));
    int64_t* flag = flag_b.as<int64_t>();
    int64_t* lidx = lidx_b.as<int64_t>();
    MI_LAUNCH(k_kp_flag, dim3((unsigned)ceil_div(m.rows, 256)), dim3(256), c.stream, (const int64_t*)m.ptr, m.rows, min_row,
              flag, slen_b.as<int64_t>());
    const int64_t n_long = exclusive_scan_i64(flag, lidx, m.rows);
    auto kp = std::make_shared<SpmmKpart>();
    kp->P = P;
    kp->chunk = o.spmm_kpart_chunk;
    kp->min_row = min_row;
    kp->n_long = n_long;
    Csr& sh = kp->shrt;
    sh.rows = m.rows;
    sh.cols = m.cols;
    sh.ptr_own.alloc(sizeof(int64_t) * (size_t)(m.rows + 1));
    sh.ptr = sh.ptr_own.as<int64_t>();
    sh.nnz = exclusive_scan_i64(slen_b.as<int64_t>(), sh.ptr, m.rows);
    kp->nnz_long = m.nnz - sh.nnz;
    // worth it when the long rows carry a real share of the gather (their partial rows cost 2 P row widths each, so a row
    // must gather a multiple of that) and there is something left to balance across the chip
    const bool pays = n_long > 0 && kp->nnz_long * 4 >= m.nnz;
    if (!(pays || (o.spmm_kpart == 2 && n_long > 0))) return;
    kp->rowid.alloc(sizeof(int32_t) * (size_t)n_long);
    MI_LAUNCH(k_kp_rowid, dim3((unsigned)ceil_div(m.rows, 256)), dim3(256), c.stream, (const int64_t*)flag, (const int64_t*)lidx,
              m.rows, kp->rowid.as<int32_t>());
    Csr& cat = kp->cat;
    cat.rows = (int64_t)P * n_long;
    cat.cols = m.cols;
    cat.ptr_own.alloc(sizeof(int64_t) * (size_t)(cat.rows + 1));
    cat.ptr = cat.ptr_own.as<int64_t>();
    DevBuf cnt_b;
    cnt_b.alloc(sizeof(int64_t) * (size_t)(cat.rows + 1));
    MI_LAUNCH(k_kp_count, dim3((unsigned)ceil_div(n_long * WAVE, 256)), dim3(256), c.stream, (const int64_t*)m.ptr,
              (const int32_t*)m.col, (const int32_t*)kp->rowid.as<int32_t>(), n_long, P, cnt_b.as<int64_t>());
    cat.nnz = exclusive_scan_i64(cnt_b.as<int64_t>(), cat.ptr, cat.rows);
    if (cat.nnz != kp->nnz_long) fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "column-partitioned plan: entry counts disagree");
    cat.col_own.alloc(sizeof(int32_t) * (size_t)cat.nnz);
    cat.val_own.alloc(vb * (size_t)cat.nnz);
    cat.col = cat.col_own.as<int32_t>();
    cat.val = cat.val_own.p;
    sh.col_own.alloc(sizeof(int32_t) * (size_t)sh.nnz);
    sh.val_own.alloc(vb * (size_t)sh.nnz);
    sh.col = sh.col_own.as<int32_t>();
    sh.val = sh.val_own.p;
    auto fill = [&](auto word) {
        using W = decltype(word);
        MI_LAUNCH((k_kp_fill_long<W>), dim3((unsigned)ceil_div(n_long * WAVE, 256)), dim3(256), c.stream, (const int64_t*)m.ptr,
                  (const int32_t*)m.col, (const W*)m.val, (const int32_t*)kp->rowid.as<int32_t>(), n_long, P,
                  (const int64_t*)cat.ptr, cat.col, (W*)cat.val);
        if (sh.nnz)
            MI_LAUNCH((k_kp_fill_short<W>), dim3((unsigned)ceil_div(m.rows * 8, 256)), dim3(256), c.stream, (const int64_t*)m.ptr,
                      (const int32_t*)m.col, (const W*)m.val, m.rows, min_row, (const int64_t*)sh.ptr, sh.col, (W*)sh.val);
    };
    if (vb == 4) fill(uint32_t{});
    else if (vb == 8) fill(uint64_t{});
    else fill(KpWord16{});
    cat.valid = sh.valid = true;
    cat.sorted = sh.sorted = false;
    // chunk ranges of the partitions: partition q starts at item cat.ptr[q n_long] + q n_long of cat's item sequence
    {
        std::vector<int64_t> starts((size_t)P + 1, 0);
        for (int q = 1; q < P; ++q)
            MI_HIP_CHECK(hipMemcpyAsync(&starts[(size_t)q], cat.ptr + (int64_t)q * n_long, sizeof(int64_t), hipMemcpyDeviceToHost,
                                        c.stream));
        MI_HIP_CHECK(hipEventRecord(e1, c.stream));
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        const int64_t chunk = kp->chunk;
        const int64_t nchunks = ceil_div(cat.nnz + cat.rows, chunk);
        kp->cs[0] = 0;
        for (int q = 1; q < P; ++q) kp->cs[q] = (starts[(size_t)q] + (int64_t)q * n_long) / chunk;
        for (int q = P; q <= 8; ++q) kp->cs[q] = nchunks;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    counters().spmm_kpart_build_ms += ms;
    p.kpart = std::move(kp);
    p.kpart_state = 2;
}

// The partitioned form is a second copy of the matrix ((4 + sizeof value) bytes per entry and a few row-sized arrays): on a
// device that cannot hold it the product must not fail -- the row-owned kernel of the first two calls still works.  The plan
// is then declined for good (kpart_state 1; set_values / order look again).
static void build_kpart(SpmmPlan& p, const Csr& m, char vtype)
{
    try {
        build_kpart_impl(p, m, vtype);
    } catch (const status_error& e) {
        if (e.status != MI_SPARSE_STATUS_ALLOC_FAILED) throw;
        clear_error();
        (void)hipGetLastError();
        p.kpart.reset();
        p.kpart_state = 1;
    }
}

// One product with plan `p` of matrix `m` (row-major or column-major strides as given; see spmm_device for the layouts).
// `parts`: column-partitioned launch of a concatenated matrix (slices = 8 / parts->P), else the plain mapping.
template <typename T>
static void spmm_run(SpmmPlan& p, std::mutex& mtx, const Csr& m, int conj_a, T alpha, int layout, const T* B, int64_t N,
                     int64_t ldb, T beta, T* C, int64_t ldc, const SpmmKpart* parts, bool hold_hot, bool allow_hot = true,
                     int chunk = 0)
{
    Context& c = ctx();
    // XCD-affine column slices (k_spmm): each set of XCDs works on N / S dense columns only
    constexpr int V16 = 16 / (int)sizeof(T);
    int slices = 1;
    {
        int64_t want = options().spmm_slices;
        if (parts) want = 8 / parts->P;
        else if (want == 0) {
            // by row width only (never by the matrix: the column split fixes the summation order, so a handle
            // returns the same bits on every call): 256-byte slices, 128-byte ones for 256-byte rows.  Measured on
            // the headline matrix (profiles/r02_spmm_slices_pmc.jsonl): 512-byte rows 2.00 -> 1.79 ms with 2 slices,
            // 1 KiB rows 4.89 -> 4.18 ms with 4; 64-byte slices double the L2 requests and lose.
            const int64_t row_bytes = N * (int64_t)sizeof(T);
            want = row_bytes >= 2048 ? 8 : row_bytes >= 1024 ? 4 : row_bytes >= 256 ? 2 : 1;
        }
        if (layout == MI_SPARSE_LAYOUT_ROW_MAJOR && (want == 2 || want == 4 || want == 8) && N % want == 0 &&
            (N / want) % V16 == 0 && (N / want) * (int64_t)sizeof(T) >= 64)
            slices = (int)want;
    }
    const int64_t slice_bytes = N / slices * (int64_t)sizeof(T);  // bytes of one B row one XCD's L2 sees
    // hot-set budget in rows of B for this call's row width (row-major operands only); the partitioned launch gathers
    // untagged (every XCD's reference stream is an eighth of the columns: non-temporal cold loads cost more than the
    // evictions they prevent -- long pass 0.81 ms untagged, 0.94-1.04 ms tagged, profiles/r05_spmm_kpart_probe.log)
    int64_t hot_rows = 0;
    if (!parts && allow_hot && layout == MI_SPARSE_LAYOUT_ROW_MAJOR && options().spmm_hot_kb > 0 &&
        (N * (int64_t)sizeof(T) >= 512 || options().spmm_hot_force))
        hot_rows = options().spmm_hot_kb * 1024 / slice_bytes;
    const SpmmPlan& pl = get_plan(p, mtx, m, chunk > 0 ? chunk : (int)options().spmm_chunk, hot_rows);
    // fix-up grid: the exact task count once it has reached the host, else its upper bound
    const int64_t fix_tasks = pl.n_tasks >= 0 ? ceil_div(pl.n_tasks, 8) + pl.n_tasks_long : pl.nchunks;  // workgroups of the fix-up
    // one workgroup per task, grid-stride beyond the grid.  While the exact count is still on its way to the host the bound is
    // the chunk count -- hundreds of thousands of workgroups that would read the count and exit on exactly the first calls:
    // a few workgroups per CU stride over whatever the count turns out to be
    int64_t fix_grid = fix_tasks < ((int64_t)1 << 20) ? fix_tasks : ((int64_t)1 << 20);
    if (pl.n_tasks < 0 && c.cus > 0 && fix_grid > (int64_t)8 * c.cus) fix_grid = (int64_t)8 * c.cus;
    const unsigned long long* n_tasks_dev = static_cast<const unsigned long long*>(pl.n_tasks_dev.p);
    T* carry_val = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)pl.nchunks * (size_t)N));
    const bool row_major = (layout == MI_SPARSE_LAYOUT_ROW_MAJOR);
    const int64_t b_rs = row_major ? ldb : 1, b_cs = row_major ? 1 : ldb;
    const int64_t c_rs = row_major ? ldc : 1, c_cs = row_major ? 1 : ldc;
    const bool vec_ok = row_major && !options().spmm_force_generic && (N % V16 == 0) &&
                        ((ldb * (int64_t)sizeof(T)) % 16 == 0) && ((ldc * (int64_t)sizeof(T)) % 16 == 0) &&
                        ((reinterpret_cast<uintptr_t>(B) % 16) == 0) && ((reinterpret_cast<uintptr_t>(C) % 16) == 0) &&
                        ((reinterpret_cast<uintptr_t>(carry_val) % 16) == 0);
    if (N == 1 && !options().spmm_force_generic) {
        // SpMV: lanes over nonzeros (k_spmv); same plan, carries and fix-up as the wide kernel
        const size_t pw = ((size_t)(pl.chunk + SPMM_SPLIT) * sizeof(T) + (size_t)(pl.chunk + 2) * sizeof(int32_t) + 15) & ~size_t(15);
        counters().spmm_last_tagged = 0.0;
        note_kernel("mi::k_spmv<%s>", type_name<T>());
        MI_LAUNCH_SMEM((k_spmv<T>), dim3((unsigned)ceil_div(pl.nchunks, SPMM_WAVES)), dim3(SPMM_WAVES * WAVE),
                       pw * SPMM_WAVES, c.stream, m.rows, m.nnz, (const int64_t*)m.ptr, (const int32_t*)m.col,
                       (const T*)m.val, (const SpmmChunk*)pl.chunk_desc.as<SpmmChunk>(), pl.nchunks, pl.chunk, conj_a, B, b_rs,
                       C, c_rs, alpha, beta, (int)(vt<T>::is_zero(beta) ? 1 : 0), carry_val);
        if (fix_tasks)
            MI_LAUNCH((k_spmm_fixup<T, 1, 16>), dim3((unsigned)fix_grid), dim3(8 * 16), c.stream,
                      n_tasks_dev, (const int32_t*)pl.tasks.as<int32_t>(), pl.nchunks, (const T*)carry_val, N, C, c_rs, c_cs, alpha);
        plan_after_product(p, mtx, m, hot_rows, hold_hot);
        return;
    }
    // tagged (hot / cold) gather: raw buffer loads, i.e. 32-bit byte offsets must reach all of B
    const int tag_mode = (pl.tagged && vec_ok && dense_bytes_below_4g(m.cols, ldb, sizeof(T))) ? SPMM_TAG_BUFFER : SPMM_TAG_NONE;
    counters().spmm_last_tagged = (double)tag_mode;
    counters().spmm_hot_coverage = pl.hot_coverage;
    if (!vec_ok) slices = 1;
    if (parts && slices != 8 / parts->P) fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "column-partitioned launch on an unaligned operand");
    counters().spmm_last_slices = (double)slices;
    SpmmParts kparts;
    kparts.P = parts ? parts->P : 0;
    for (int q = 0; q < 9; ++q) kparts.cs[q] = parts ? parts->cs[q] : 0;
#define MI_SPMM_ARGS m, pl, conj_a, B, b_rs, b_cs, C, c_rs, c_cs, N, alpha, beta, carry_val, tag_mode, slices, kparts
    if (vec_ok) {
        const int64_t lanes = N / slices / V16;  // 16-byte lanes needed for one (slice of a) row of B
        if (lanes >= 64) launch_spmm<T, V16, 64>(MI_SPMM_ARGS);
        else if (lanes > 16) launch_spmm<T, V16, 32>(MI_SPMM_ARGS);
        else if (lanes > 8) launch_spmm<T, V16, 16>(MI_SPMM_ARGS);
        else if (lanes > 4 || slices == 1) launch_spmm<T, V16, 8>(MI_SPMM_ARGS);
        else launch_spmm<T, V16, 4>(MI_SPMM_ARGS);
    } else {
        if (N > 16) launch_spmm<T, 1, 64>(MI_SPMM_ARGS);
        else if (N > 4) launch_spmm<T, 1, 16>(MI_SPMM_ARGS);
        else launch_spmm<T, 1, 4>(MI_SPMM_ARGS);
    }
#undef MI_SPMM_ARGS
    if (fix_tasks) {
        const int32_t* tk = pl.tasks.as<int32_t>();
        if (vec_ok) {
            if (N / V16 > 16)
                MI_LAUNCH((k_spmm_fixup<T, V16, 32>), dim3((unsigned)fix_grid), dim3(8 * 32), c.stream,
                          n_tasks_dev, tk, pl.nchunks, (const T*)carry_val, N, C, c_rs, c_cs, alpha);
            else
                MI_LAUNCH((k_spmm_fixup<T, V16, 8>), dim3((unsigned)fix_grid), dim3(8 * 8), c.stream,
                          n_tasks_dev, tk, pl.nchunks, (const T*)carry_val, N, C, c_rs, c_cs, alpha);
        } else {
            MI_LAUNCH((k_spmm_fixup<T, 1, 16>), dim3((unsigned)fix_grid), dim3(8 * 16), c.stream,
                      n_tasks_dev, tk, pl.nchunks, (const T*)carry_val, N, C, c_rs, c_cs, alpha);
        }
    }
    plan_after_product(p, mtx, m, hot_rows, hold_hot);
}

// Core executor on DEVICE pointers.  m is the CSR of op(A) (rows of m = rows of C).
template <typename T>
void spmm_device(mi_sparse_matrix* h, bool transposed, const Csr& m, int conj_a, T alpha, int layout, const T* B,
                 int64_t N, int64_t ldb, T beta, T* C, int64_t ldc)
{
    Context& c = ctx();
    if (m.rows == 0 || N == 0) return;
    if (layout == MI_SPARSE_LAYOUT_COLUMN_MAJOR && N > 1 && !options().spmm_force_generic) {
        // Column-major operands: a gather of B "rows" would touch one element per cache line.
        // Re-lay B (and C when beta != 0) as row-major scratch copies with a 16-byte-aligned
        // leading dimension, run the coalesced kernel, and write C back column-major: two extra
        // streaming passes over the dense operands instead of an N-fold amplified gather.
        constexpr int64_t A16 = 16 / (int64_t)sizeof(T);
        const int64_t ldt = ceil_div(N, A16) * A16;
        T* bt = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)m.cols * (size_t)ldt));
        T* ct = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)m.rows * (size_t)ldt));
        convert_layout<T>(m.cols, N, B, 1, ldb, bt, ldt, 1);
        if (!vt<T>::is_zero(beta)) convert_layout<T>(m.rows, N, C, 1, ldc, ct, ldt, 1);
        spmm_device<T>(h, transposed, m, conj_a, alpha, MI_SPARSE_LAYOUT_ROW_MAJOR, bt, N, ldt, beta, ct, ldt);
        convert_layout<T>(m.rows, N, ct, ldt, 1, C, 1, ldc);
        return;
    }
    // option profile_events: hipEvents on the launch stream around EVERY kernel of the product (main kernels, carry fix-ups,
    // the combine of a partitioned product) -- counter spmm_kernel_ms / spmm_kernel_launches = device time per product
    struct ProductTimer {
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        hipStream_t s;
        explicit ProductTimer(hipStream_t st) : s(st)
        {
            if (!options().profile_events) return;
            if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) return;
            (void)hipEventRecord(ev0, s);
        }
        ~ProductTimer()
        {
            if (ev0 && ev1 && hipEventRecord(ev1, s) == hipSuccess && hipEventSynchronize(ev1) == hipSuccess) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) {
                    counters().spmm_kernel_ms += ms;
                    counters().spmm_kernel_launches += 1.0;
                }
            }
            if (ev0) (void)hipEventDestroy(ev0);
            if (ev1) (void)hipEventDestroy(ev1);
        }
    } product_timer(c.stream);
    SpmmPlan& p = transposed ? h->planT : h->plan;
    // Column-partitioned long rows (SpmmKpart): for operands the vector path takes, rows of B of at least 256 bytes, a B
    // beyond what the L2s hold between them, and a matrix large enough for the split to matter.  Looked at ahead of the
    // THIRD product of a handle (two products prove the reuse; a single-use handle never pays for the build).  The
    // partial sums of a partitioned row are added in a fixed order, so results stay bitwise reproducible call to call --
    // but they are not the bits of the row-owned product (another summation order): option deterministic keeps the
    // row-owned kernel for good.
    constexpr int V16 = 16 / (int)sizeof(T);
    const Options& o = options();
    const int kp_slices = 8 / (int)o.spmm_kpart_parts;
    const bool kp_shape = o.spmm_kpart != 0 && !o.deterministic && !o.spmm_force_generic && N > 1 &&
                          layout == MI_SPARSE_LAYOUT_ROW_MAJOR && N % (V16 * kp_slices) == 0 &&
                          (N / kp_slices) * (int64_t)sizeof(T) >= 64 && (ldb * (int64_t)sizeof(T)) % 16 == 0 &&
                          (ldc * (int64_t)sizeof(T)) % 16 == 0 && (reinterpret_cast<uintptr_t>(B) % 16) == 0 &&
                          (reinterpret_cast<uintptr_t>(C) % 16) == 0 &&
                          (o.spmm_kpart == 2 || (N * (int64_t)sizeof(T) >= 256 && m.nnz >= ((int64_t)1 << 21) &&
                                                 (double)m.cols * (double)N * (double)sizeof(T) >= 64.0 * 1048576.0));
    bool hold_hot = false;
    std::shared_ptr<SpmmKpart> kp_keep;  // this call's reference: another host thread may drop the plan (set_values) meanwhile
    if (kp_shape) {
        std::lock_guard<std::mutex> lk(h->mtx);
        if (p.kpart_state == 2 && (p.kpart->P != (int)o.spmm_kpart_parts || p.kpart->min_row != o.spmm_kpart_min_row ||
                                   p.kpart->chunk != o.spmm_kpart_chunk))
            p.kpart_state = 0;  // the options changed (tools; the partitions' chunk ranges follow spmm_kpart_chunk): build again
        if (p.kpart_state == 0 && (p.uses >= 2 || o.spmm_plan_sync || o.spmm_kpart == 2)) build_kpart(p, m, h->vtype);
        hold_hot = p.kpart_state != 1;
        if (p.kpart_state == 2) kp_keep = p.kpart;
    }
    // the partial rows (one per long row and partition) live in the scratch arena: not for operands so wide that they would
    // take a real share of the device (its size: asked once per host thread, outside the handle's mutex)
    if (kp_keep && (double)kp_keep->cat.rows * (double)N * (double)sizeof(T) > (double)device_total_bytes() / 8.0) kp_keep.reset();
    if (kp_keep) {
        SpmmKpart& kp = *kp_keep;
        counters().spmm_last_kpart = (double)kp.P;
        counters().spmm_kpart_long_share = m.nnz ? (double)kp.nnz_long / (double)m.nnz : 0.0;
        // short rows, row-owned, straight into C (the long rows are empty there: they get beta * C).  Gathered UNTAGGED: once the
        // long rows are gone, what is left re-references too little for the slower non-temporal loads of the cold columns to pay
        // (headline 1.30 -> 1.25 ms, three interleaved rounds; 512-byte rows of B; 1 KiB rows: 2.99 vs 3.02 ms --
        // profiles/r05_spmm_short_rows_untagged_ab.log); the row-owned product of the WHOLE matrix keeps its tags (1.72 vs 1.88 ms)
        spmm_run<T>(kp.plan_short, h->mtx, kp.shrt, conj_a, alpha, layout, B, N, ldb, beta, C, ldc, nullptr, true, false,
                    (int)kp.chunk);
        // long rows: partial[q * n_long + i] = (sub-row q of long row i) * B, then C[rowid[i]] += alpha * sum over q
        T* partial = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)kp.cat.rows * (size_t)N));
        spmm_run<T>(kp.plan_cat, h->mtx, kp.cat, conj_a, vt<T>::one(), layout, B, N, ldb, vt<T>::zero(), partial, N, &kp, true,
                    true, (int)kp.chunk);
        const int64_t lanes = N / V16;
        if (lanes > 16)
            MI_LAUNCH((k_kp_combine<T, V16, 32>), dim3((unsigned)ceil_div(kp.n_long * 32, 256)), dim3(256), c.stream,
                      (const T*)partial, kp.n_long, kp.P, (const int32_t*)kp.rowid.as<int32_t>(), N, C, ldc, alpha,
                      (int)(vt<T>::is_zero(beta) ? 1 : 0));
        else
            MI_LAUNCH((k_kp_combine<T, V16, 8>), dim3((unsigned)ceil_div(kp.n_long * 8, 256)), dim3(256), c.stream,
                      (const T*)partial, kp.n_long, kp.P, (const int32_t*)kp.rowid.as<int32_t>(), N, C, ldc, alpha,
                      (int)(vt<T>::is_zero(beta) ? 1 : 0));
        std::lock_guard<std::mutex> lk(h->mtx);
        ++p.uses;
        return;
    }
    counters().spmm_last_kpart = 0.0;
    spmm_run<T>(p, h->mtx, m, conj_a, alpha, layout, B, N, ldb, beta, C, ldc, nullptr, hold_hot);
}

template void spmm_device<float>(mi_sparse_matrix*, bool, const Csr&, int, float, int, const float*, int64_t, int64_t,
                                 float, float*, int64_t);
template void spmm_device<double>(mi_sparse_matrix*, bool, const Csr&, int, double, int, const double*, int64_t,
                                  int64_t, double, double*, int64_t);
template void spmm_device<cfloat>(mi_sparse_matrix*, bool, const Csr&, int, cfloat, int, const cfloat*, int64_t,
                                  int64_t, cfloat, cfloat*, int64_t);
template void spmm_device<cdouble>(mi_sparse_matrix*, bool, const Csr&, int, cdouble, int, const cdouble*, int64_t,
                                   int64_t, cdouble, cdouble*, int64_t);

static size_t dense_extent(int layout, int64_t r, int64_t cdim, int64_t ld)
{
    if (r == 0 || cdim == 0) return 0;
    return (layout == MI_SPARSE_LAYOUT_ROW_MAJOR) ? (size_t)((r - 1) * ld + cdim) : (size_t)((cdim - 1) * ld + r);
}

template <typename T>
static int mm_generic(int op, T alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr, int layout, const T* B,
                      int64_t columns, int64_t ldb, T beta, T* C, int64_t ldc)
{
    return guarded([&] {
        mi_sparse_matrix* h = check_handle(A);
        if (h->vtype != type_char<T>::value)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "handle holds '%c' values but the '%c' routine was called", h->vtype,
                 type_char<T>::value);
        if (descr.type != MI_SPARSE_MATRIX_TYPE_GENERAL)
            fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "only SPARSE_MATRIX_TYPE_GENERAL descriptors are supported");
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE && op != MI_SPARSE_OPERATION_TRANSPOSE &&
            op != MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad operation code %d", op);
        if (layout != MI_SPARSE_LAYOUT_ROW_MAJOR && layout != MI_SPARSE_LAYOUT_COLUMN_MAJOR)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad layout code %d", layout);
        if (columns < 0) fail(MI_SPARSE_STATUS_INVALID_VALUE, "negative column count");
        const bool trans = (op != MI_SPARSE_OPERATION_NON_TRANSPOSE);
        const int64_t crows = trans ? h->cols : h->rows;
        const int64_t brows = trans ? h->rows : h->cols;
        const bool row_major = (layout == MI_SPARSE_LAYOUT_ROW_MAJOR);
        if (ldb < (row_major ? columns : brows) || ldc < (row_major ? columns : crows))
            if (columns > 0 && crows > 0 && brows > 0)
                fail(MI_SPARSE_STATUS_INVALID_VALUE, "leading dimension too small (ldb=%lld ldc=%lld)", (long long)ldb,
                     (long long)ldc);
        if (crows == 0 || columns == 0) return;
        if (!C || (!B && brows > 0)) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL dense operand");
        Context& c = ctx();
        c.scratch_reset();
        const int conj_a = (op == MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE && vt<T>::is_complex) ? 1 : 0;
        Staged sb, sc;
        sb.stage_in(B, sizeof(T) * dense_extent(layout, brows, columns, ldb), true);
        sc.stage_in(C, sizeof(T) * dense_extent(layout, crows, columns, ldc), !vt<T>::is_zero(beta));
        if (!trans && bsr_spmm_applicable<T>(h->bsr, layout, static_cast<const T*>(sb.dev), columns, ldb,
                                             static_cast<const T*>(sc.dev), ldc)) {
            // handle created from BSR arrays: the block kernel on the block form (no CSR index traffic)
            bsr_spmm_device<T>(h->bsr, alpha, static_cast<const T*>(sb.dev), columns, ldb, beta, static_cast<T*>(sc.dev), ldc);
        } else {
            Csr& m = trans ? need_csrT(h) : need_csr(h);
            spmm_device<T>(h, trans, m, conj_a, alpha, layout, static_cast<const T*>(sb.dev), columns, ldb, beta,
                           static_cast<T*>(sc.dev), ldc);
        }
        MI_HIP_CHECK(hipGetLastError());
        if (sb.host) c.sync();  // staged B is freed when `sb` goes out of scope
        sc.copy_back();
    });
}

}  // namespace mi

using mi::cdouble;
using mi::cfloat;

static inline cfloat cv(mi_complex8 x) { return cfloat{x.real, x.imag}; }
static inline cdouble cv(mi_complex16 x) { return cdouble{x.real, x.imag}; }

extern "C" {

/* The inspector stage (the analogue of mkl_sparse_optimize, which the reference never calls: it creates a handle per product,
 * _common.py:245-293): everything the library would otherwise learn about A over its first three products as a left
 * operand, done now -- the row partition and fix-up schedule, and for matrices with long rows the column-partitioned
 * form (SpmmKpart).  The next product runs the steady-state kernels.  Optional: products are correct without it. */
mi_sparse_status_t mi_sparse_optimize(mi_sparse_matrix_t A)
{
    return mi::guarded([&] {
        mi_sparse_matrix* h = mi::check_handle(A);
        mi::Context& c = mi::ctx();
        c.scratch_reset();
        const mi::Options& o = mi::options();
        mi::Csr& m = mi::need_csr(h);
        {
            std::lock_guard<std::mutex> lk(h->mtx);
            mi::SpmmPlan& p = h->plan;
            if (p.uses < 2) p.uses = 2;  // reuse is declared, not observed: the hot / cold analysis follows the next product
            if (o.spmm_kpart != 0 && !o.deterministic && !o.spmm_force_generic && p.kpart_state == 0 &&
                (o.spmm_kpart == 2 || m.nnz >= ((int64_t)1 << 21)))
                mi::build_kpart(p, m, h->vtype);
        }
        c.sync();
    });
}

mi_sparse_status_t mi_sparse_s_mm(int op, float alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr, int layout,
                                  const float* B, int64_t columns, int64_t ldb, float beta, float* C, int64_t ldc)
{
    return mi::mm_generic<float>(op, alpha, A, descr, layout, B, columns, ldb, beta, C, ldc);
}
mi_sparse_status_t mi_sparse_d_mm(int op, double alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr, int layout,
                                  const double* B, int64_t columns, int64_t ldb, double beta, double* C, int64_t ldc)
{
    return mi::mm_generic<double>(op, alpha, A, descr, layout, B, columns, ldb, beta, C, ldc);
}
mi_sparse_status_t mi_sparse_c_mm(int op, mi_complex8 alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  int layout, const mi_complex8* B, int64_t columns, int64_t ldb, mi_complex8 beta,
                                  mi_complex8* C, int64_t ldc)
{
    return mi::mm_generic<cfloat>(op, cv(alpha), A, descr, layout, (const cfloat*)B, columns, ldb, cv(beta),
                                  (cfloat*)C, ldc);
}
mi_sparse_status_t mi_sparse_z_mm(int op, mi_complex16 alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  int layout, const mi_complex16* B, int64_t columns, int64_t ldb, mi_complex16 beta,
                                  mi_complex16* C, int64_t ldc)
{
    return mi::mm_generic<cdouble>(op, cv(alpha), A, descr, layout, (const cdouble*)B, columns, ldb, cv(beta),
                                   (cdouble*)C, ldc);
}

mi_sparse_status_t mi_sparse_s_mv(int op, float alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const float* x, float beta, float* y)
{
    return mi::mm_generic<float>(op, alpha, A, descr, MI_SPARSE_LAYOUT_ROW_MAJOR, x, 1, 1, beta, y, 1);
}
mi_sparse_status_t mi_sparse_d_mv(int op, double alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const double* x, double beta, double* y)
{
    return mi::mm_generic<double>(op, alpha, A, descr, MI_SPARSE_LAYOUT_ROW_MAJOR, x, 1, 1, beta, y, 1);
}
mi_sparse_status_t mi_sparse_c_mv(int op, mi_complex8 alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const mi_complex8* x, mi_complex8 beta, mi_complex8* y)
{
    return mi::mm_generic<cfloat>(op, cv(alpha), A, descr, MI_SPARSE_LAYOUT_ROW_MAJOR, (const cfloat*)x, 1, 1,
                                  cv(beta), (cfloat*)y, 1);
}
mi_sparse_status_t mi_sparse_z_mv(int op, mi_complex16 alpha, mi_sparse_matrix_t A, struct mi_matrix_descr descr,
                                  const mi_complex16* x, mi_complex16 beta, mi_complex16* y)
{
    return mi::mm_generic<cdouble>(op, cv(alpha), A, descr, MI_SPARSE_LAYOUT_ROW_MAJOR, (const cdouble*)x, 1, 1,
                                   cv(beta), (cdouble*)y, 1);
}

}  // extern "C"
