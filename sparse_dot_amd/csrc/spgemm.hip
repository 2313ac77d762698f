// spgemm.hip -- sparse x sparse products:
//   mi_sparse_spmm      C := A * B            (sparse CSR out)   reference _sparse_sparse.py:35-40
//   mi_sparse_?_spmmd   C := A * B            (dense out)        reference _sparse_sparse.py:94-101
//   mi_sparse_syrk      C := triu(A^T A) / triu(A A^T) (sparse)  reference _gram_matrix.py:70-74
//
// Two-phase row-wise (Gustavson) SpGEMM with hash accumulators:
//   0. ub[i]  = sum over nonzeros (i,k) of A of nnz(B[k,:])          (upper bound of row i of C)
//   1. symbolic: rows binned by ub; every row counts its distinct columns in a hash table
//      (LDS tables of 64 / 256 / 1024 / 4096 slots per workgroup for ub <= 32 / 128 / 512 / 2048; a global-memory slab per persistent workgroup
//      for the hub rows of skewed matrices)
//   2. exclusive scan of the counts -> row pointer of C (int64: nnz(C) may exceed 2^31)
//   3. numeric: rows re-binned by their exact length; same hash tables now carry values
//      (LDS float/double atomic adds), then the table is compacted into C.
// Column order inside a row of C is unspecified (as with mkl_sparse_spmm); mi_sparse_order sorts.
// Entries that cancel to 0.0 stay (MKL keeps them; scipy prunes -- SURVEY section 8 a3).
#include "common.hpp"
#include <chrono>
#include <memory>

namespace mi {

constexpr int32_t HASH_EMPTY = -1;

// how a numeric kernel writes a column index of C: key + base (B is a column panel of a wider matrix, spgemm_panels).
// (Round 6 also carried an optional relabelling map here for the hub path: the `map ? map[key] : key + base` in every write-out
// cost the uniform configs[2] 13 % -- 2.87 -> 3.25 ms -- with the map unused; the hub path maps its columns back in a pass of its own.)
struct ColOut {
    int32_t base = 0;
    __device__ __forceinline__ int32_t operator()(int32_t key) const { return key + base; }
};

__device__ __forceinline__ uint32_t hash_col(int32_t c, int log2s)
{
    return (uint32_t)((uint32_t)c * 2654435761u) >> (32 - log2s);
}

__device__ __forceinline__ int64_t lower_bound_col(const int32_t* __restrict__ col, int64_t lo, int64_t hi, int32_t key)
{
    while (lo < hi) {  // first position in [lo, hi) with col >= key
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---- phase 0: upper bounds -----------------------------------------------------------------------
// `upper`: 0 = full product, 1 = upper triangle only (products with column < row are dropped one by one),
// 2 = upper triangle and the rows of B are sorted (the dropped part of every B row is skipped by a search).
// Round 3: the extent of B's row for every nonzero of A -- (first entry that counts, number of entries), the lower
// triangle already cut off -- is WRITTEN OUT here (ext0 / extlen, 12 bytes per nonzero of A, streamed) and read back by the
// symbolic and numeric kernels.  bptr[k] for a random k is a 128-byte line fill for 16 useful bytes, and it was gathered
// three times (here, symbolic, numeric): with 16-entry rows of B that was one line in 2.5 (symbolic) / 4.5 (numeric).
// Round 4: BP = int32_t reads a narrowed copy of B's row pointer (k_narrow_ptr, scratch of the call).  The gather of bptr[k] is a
// 128-byte line per nonzero of A wherever it misses, and the 8 MB pointer of a 2^20-row B is twice an XCD's L2: on the uniform
// configs[2] the kernel fetched 1.5 GB for 67 MB of column indices (L2 hit 0.42, profiles/r04_pmc_spgemm_uniform_kernels.jsonl).
__global__ void k_narrow_ptr(const int64_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (int32_t)src[i];
}

constexpr int ROWUB_LPR_LOG2 = 4;    // lanes per row of A
constexpr int ROWUB_LONG = 512;      // rows of A with more nonzeros go to k_row_ub_long (a workgroup each)

__device__ __forceinline__ void row_ub_one(int64_t p, int32_t row, const int32_t* __restrict__ acol, int64_t b0, int64_t b1,
                                           const int32_t* __restrict__ bcol, int upper, int64_t& s, int64_t* __restrict__ ext0,
                                           int32_t* __restrict__ extlen, int32_t row_shift)
{
    // row_shift: B is a column PANEL of a wider matrix with its columns rebased (spgemm_panels): the diagonal sits at column row - shift
    if (upper == 2 && b0 < b1) b0 = lower_bound_col(bcol, b0, b1, row - row_shift);
    s += b1 - b0;
    ext0[p] = b0;
    extlen[p] = (int32_t)(b1 - b0);
}

// 16 lanes per row; rows with more than ROWUB_LONG nonzeros are only LISTED here.  Round 4: with 8 lanes per row and one
// dependent acol -> bptr chain per step the hub rows of A set the kernel's duration -- 0.96 ms on R-MAT 2^18 and 3.0 ms on the
// literal configs[2] for 17 M nonzeros, against 0.18 ms for the uniform matrix of the same size.
template <typename BP>
__global__ void __launch_bounds__(256)
    k_row_ub(int64_t rows, const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol,
             const BP* __restrict__ bptr, const int32_t* __restrict__ bcol, int upper, int64_t* __restrict__ ub,
             int64_t* __restrict__ ext0, int32_t* __restrict__ extlen, int32_t* __restrict__ long_list,
             unsigned* __restrict__ long_count, int32_t row_shift)
{
    constexpr int LPR = 1 << ROWUB_LPR_LOG2;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = t >> ROWUB_LPR_LOG2;
    const int sub = (int)(t & (LPR - 1));
    int64_t s = 0;
    bool is_long = false;
    if (row < rows) {
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        is_long = a1 - a0 > ROWUB_LONG;
        if (is_long) {
            if (sub == 0) long_list[atomicAdd(long_count, 1u)] = (int32_t)row;
        } else if (upper != 2) {
            for (int64_t p = a0 + sub; p < a1; p += LPR) {
                const int32_t k = acol[p];
                row_ub_one(p, (int32_t)row, acol, (int64_t)bptr[k], (int64_t)bptr[k + 1], bcol, upper, s, ext0, extlen, row_shift);
            }
        }
    }
    if (upper == 2) {
        // Upper triangle, sorted rows of B: where row k of B crosses the diagonal.  A bisection per nonzero of A is four or five
        // DEPENDENT gathers (1.08 of the 3.8 ms of a sparse gram product with 16-entry rows).  Rows of B of at most 16 entries
        // are instead read whole by the 16 lanes of the group -- sixteen nonzeros of A per step, their sixteen loads issued
        // together, the position = the number of entries left of the diagonal (one ballot each); longer rows keep the search.
        // (Every lane of the wave runs the loop: rows past the end / listed rows have an empty extent.)
        const bool mine = row < rows && !is_long;
        const int64_t a0 = mine ? aptr[row] : 0, a1 = mine ? aptr[row + 1] : 0;
        const int grp_shift = (threadIdx.x & 63) & ~(LPR - 1);
        const int32_t diag = (int32_t)row - row_shift;
        int64_t steps = (a1 - a0 + LPR - 1) / LPR;
#pragma unroll
        for (int d = LPR; d < 64; d <<= 1) {  // the longest row of the wave's four groups
            const int64_t o = __shfl_xor(steps, d);
            steps = o > steps ? o : steps;
        }
        for (int64_t it = 0; it < steps; ++it) {
            const int64_t p = a0 + it * LPR + sub;
            const bool valid = p < a1;
            int64_t b0 = 0, b1 = 0;
            if (valid) {
                const int32_t k = acol[p];
                b0 = (int64_t)bptr[k];
                b1 = (int64_t)bptr[k + 1];
            }
            const int len = (int)(b1 - b0);
            int32_t cv[LPR];
#pragma unroll
            for (int e = 0; e < LPR; ++e) {
                const int64_t be = __shfl(b0, e, LPR);
                const int le = __shfl(len, e, LPR);
                cv[e] = (le <= LPR && sub < le) ? bcol[be + sub] : 0x7fffffff;
            }
            int left = 0;
#pragma unroll
            for (int e = 0; e < LPR; ++e) {
                const unsigned long long m = __ballot(cv[e] < diag);
                const int cnt = __popc((unsigned)((m >> grp_shift) & ((1u << LPR) - 1u)));
                if (sub == e) left = cnt;
            }
            if (valid) {
                int64_t start = b0;
                if (b0 < b1) start = len <= LPR ? b0 + left : lower_bound_col(bcol, b0, b1, diag);
                s += b1 - start;
                ext0[p] = start;
                extlen[p] = (int32_t)(b1 - start);
            }
        }
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) s += __shfl_xor(s, d);
    if (row < rows && sub == 0 && !is_long) ub[row] = s;
}

// the listed rows: one workgroup per row, every thread one nonzero per step (no lane waits on a chain longer than
// nnz(row) / 256 steps); the grid is fixed, the number of rows is read on the device
template <typename BP>
__global__ void __launch_bounds__(256)
    k_row_ub_long(const int32_t* __restrict__ long_list, const unsigned* __restrict__ long_count,
                  const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol, const BP* __restrict__ bptr,
                  const int32_t* __restrict__ bcol, int upper, int64_t* __restrict__ ub, int64_t* __restrict__ ext0,
                  int32_t* __restrict__ extlen, int32_t row_shift)
{
    __shared__ long long wave_s[4];
    const unsigned n = *long_count;
    for (unsigned i = blockIdx.x; i < n; i += gridDim.x) {
        const int32_t row = long_list[i];
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        int64_t s = 0;
        for (int64_t p = a0 + threadIdx.x; p < a1; p += 256) {
            const int32_t k = acol[p];
            row_ub_one(p, row, acol, (int64_t)bptr[k], (int64_t)bptr[k + 1], bcol, upper, s, ext0, extlen, row_shift);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) s += __shfl_xor(s, d);
        __syncthreads();  // the previous row's totals have been read
        if ((threadIdx.x & 63) == 0) wave_s[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) ub[row] = wave_s[0] + wave_s[1] + wave_s[2] + wave_s[3];
    }
}

// ---- binning -------------------------------------------------------------------------------------
// Bin k holds the rows with 32 * 2^(k-1) < count <= 32 * 2^k (bin 0: count <= 32; the last bin: everything
// larger).  Bins 0-6 (<= 2048) always run on LDS hash tables of twice their limit, bin 7 (<= 4096) too
// unless the values are complex double (table too large), bin 8 (<= 8192) in the symbolic phase only
// (keys fit, values do not).  Larger rows: LDS bitmap / range-partitioned hash, or the global-memory hash,
// taken largest class first (longest-processing-time-first scheduling).
constexpr int NBINS = 22;
__host__ __device__ inline int64_t bin_limit(int k) { return (int64_t)32 << k; }
__host__ __device__ inline int bin_of(int64_t c)
{
    int b = 0;
    while (c > bin_limit(b) && b < NBINS - 1) ++b;
    return b;
}

// scatters the row ids into the lists of their size classes (rows with c == 0 are skipped; NBINS <= blockDim).
// One global atomic per (workgroup, bin): positions inside the workgroup come from LDS counters.
struct BinLists {
    int32_t* l[NBINS];
};
__global__ void __launch_bounds__(256)
    k_bin_rows(int64_t rows, const int64_t* __restrict__ cnt, int64_t* __restrict__ bin_counts, BinLists lists)
{
    __shared__ int local_n[NBINS];
    __shared__ int64_t base[NBINS];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < NBINS) local_n[threadIdx.x] = 0;
    __syncthreads();
    int b = -1, pos = 0;
    if (i < rows) {
        const int64_t c = cnt[i];
        if (c > 0) {
            b = bin_of(c);
            pos = atomicAdd(&local_n[b], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < NBINS && local_n[threadIdx.x] > 0)
        base[threadIdx.x] = (int64_t)atomicAdd((unsigned long long*)&bin_counts[threadIdx.x],
                                               (unsigned long long)local_n[threadIdx.x]);
    __syncthreads();
    if (b >= 0) lists.l[b][base[b] + pos] = (int32_t)i;
}

// Inclusive prefix sums of one int per thread over a workgroup of NT threads, left in inc[0..NT): a
// shuffle scan inside each wave, then the wave totals through LDS -- two barriers instead of the
// 2 * log2(NT) of a Hillis-Steele scan (this runs once per NT nonzeros of A in the big-row kernels).
template <int NT>
__device__ __forceinline__ void block_scan_inclusive(int v, int* inc, int* wave_tot, int tid)
{
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    int add = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k)
        if (k < w) add += wave_tot[k];
    inc[tid] = v + add;
    __syncthreads();
}

template <int N>
__device__ __forceinline__ int flat_find(const int* inc, int f)
{
    int l = 0;
#pragma unroll
    for (int step = N / 2; step > 0; step >>= 1)
        if (inc[l + step - 1] <= f) l += step;
    return l;
}

// U positions at once, in LOCKSTEP: the U probes of a step are issued together and waited for together -- log2(N) LDS round
// trips for the whole batch.  Left to the compiler, U calls of flat_find run one after the other (every probe of a search
// depends on the one before and is followed by s_waitcnt lgkmcnt(0)): U * log2(N) dependent round trips per batch, about
// half of a k_spgemm_part batch's latency (round 4, from the kernel's machine code).
template <int N, int U>
__device__ __forceinline__ void flat_find_lockstep(const int* inc, const int (&f)[U], int (&l)[U])
{
#pragma unroll
    for (int u = 0; u < U; ++u) l[u] = 0;
#pragma unroll
    for (int step = N / 2; step > 0; step >>= 1) {
        int probe[U];
#pragma unroll
        for (int u = 0; u < U; ++u) probe[u] = inc[l[u] + step - 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (probe[u] <= f[u]) l[u] += step;
    }
}

// Walking the flat product list: a WAVE takes 64 * U consecutive products (lane + 64 u), so the products of one lane
// are 64 apart and mostly stay inside one slice (the B rows that matter are long): the slice of the first is found by
// bisection, the following ones by stepping forward -- about one LDS read per product instead of log2(N) + 2.
template <int N>
struct FlatCursor {
    int l, lo, hi;  // slice l holds the flat positions [lo, hi)
    __device__ __forceinline__ void seek(const int* inc, int f)
    {
        l = flat_find<N>(inc, f);
        lo = l ? inc[l - 1] : 0;
        hi = inc[l];
    }
    // f >= the previous position and < the total; true when the slice changed
    __device__ __forceinline__ bool advance(const int* inc, int f)
    {
        bool moved = false;
        while (f >= hi) {
            ++l;
            lo = hi;
            hi = inc[l];
            moved = true;
        }
        return moved;
    }
};

// ---- LDS hash kernel: one workgroup per row -------------------------------------------------------
// The B rows selected by the row of A are taken THREADS at a time: every thread fetches the extent of
// one of them, a workgroup scan turns the lengths into offsets, and the products are walked as one flat
// list -- the dependent chain acol -> bptr -> bcol is paid once per THREADS nonzeros of A, not once per
// nonzero, and a row of A with thousands of nonzeros but few surviving products (the last rows of an
// upper-triangular gram matrix) does not serialise on one lane group.
#ifndef MI_BIN3_THREADS  // tuning hooks
#define MI_BIN3_THREADS 64
#endif
#ifndef MI_BIN4_THREADS
#define MI_BIN4_THREADS 128
#endif
#ifndef MI_LDS_UNROLL
#define MI_LDS_UNROLL 4
#endif
constexpr int LDS_UNROLL = MI_LDS_UNROLL;
template <typename T, int LOG2S, int THREADS, bool NUMERIC>
__global__ void __launch_bounds__(THREADS)
    k_spgemm_lds(const int32_t* __restrict__ row_list, const int64_t* __restrict__ aptr,
                 const int64_t* __restrict__ ext0, const int32_t* __restrict__ extlen, const T* __restrict__ aval,
                 const int32_t* __restrict__ bcol, const T* __restrict__ bval, int gw, int upper,
                 int64_t* __restrict__ row_nnz, const int64_t* __restrict__ cptr, int32_t* __restrict__ ccol,
                 T* __restrict__ cval, ColOut col_base)
{
    constexpr int S = 1 << LOG2S;
    __shared__ int32_t keys[S];
    __shared__ T vals[NUMERIC ? S : 1];
    __shared__ int64_t qlo[THREADS];
    __shared__ T a_s[NUMERIC ? THREADS : 1];
    __shared__ int inc[THREADS];
    __shared__ int wave_tot[THREADS / 64];
    __shared__ int counter;
    const int tid = threadIdx.x;
    const int32_t row = row_list[blockIdx.x];
    for (int k = tid; k < S; k += THREADS) {
        keys[k] = HASH_EMPTY;
        if (NUMERIC) vals[k] = vt<T>::zero();
    }
    if (tid == 0) counter = 0;
    __syncthreads();

    int local = 0;
    const int64_t a0 = aptr[row], a1 = aptr[row + 1];
    int64_t out0 = 0;
    if (NUMERIC) out0 = cptr[row];  // needed last: issued first so that its latency is hidden
    for (int64_t base = a0; base < a1; base += THREADS) {
        int len = 0;
        if (base + tid < a1) {  // extent of B's row for this nonzero: precomputed by k_row_ub (coalesced, no gather of bptr)
            qlo[tid] = ext0[base + tid];
            len = extlen[base + tid];
            if (NUMERIC) a_s[tid] = aval[base + tid];
        }
        block_scan_inclusive<THREADS>(len, inc, wave_tot, tid);
        const int total = inc[THREADS - 1];
        for (int g0 = 0; g0 < total; g0 += THREADS * LDS_UNROLL) {
            int32_t j[LDS_UNROLL];
            T v[LDS_UNROLL];
            const int f0 = g0 + (tid >> 6) * 64 * LDS_UNROLL + (tid & 63);
            FlatCursor<THREADS> cur;
            int64_t qb = 0;
            T av = vt<T>::zero();
            if (f0 < total) {
                cur.seek(inc, f0);
                qb = qlo[cur.l];
                if (NUMERIC) av = a_s[cur.l];
            }
            // loads of B unconditional and issued together, uses after the last one (see k_spgemm_part)
            int64_t q[LDS_UNROLL];
            T avs[LDS_UNROLL];
#pragma unroll
            for (int u = 0; u < LDS_UNROLL; ++u) {
                const int f = f0 + u * 64;
                q[u] = 0;
                avs[u] = av;
                if (f < total) {
                    if (u && cur.advance(inc, f)) {
                        qb = qlo[cur.l];
                        if (NUMERIC) av = a_s[cur.l];
                    }
                    q[u] = qb + (f - cur.lo);
                    avs[u] = av;
                }
            }
#pragma unroll
            for (int u = 0; u < LDS_UNROLL; ++u) {
                j[u] = bcol[q[u]];
                if (NUMERIC) v[u] = bval[q[u]];
            }
#pragma unroll
            for (int u = 0; u < LDS_UNROLL; ++u) {
                if (f0 + u * 64 >= total) j[u] = -1;
                if (NUMERIC) v[u] = vt<T>::mul(avs[u], v[u]);
            }
#pragma unroll
            for (int u = 0; u < LDS_UNROLL; ++u) {
                if (j[u] < 0 || (upper && j[u] < row)) continue;
                uint32_t h = hash_col(j[u], LOG2S);
                for (;;) {
                    const int32_t old = atomicCAS(&keys[h], HASH_EMPTY, j[u]);
                    if (old == HASH_EMPTY || old == j[u]) {
                        if (NUMERIC) atomic_accum(&vals[h], v[u]);
                        else if (old == HASH_EMPTY) ++local;
                        break;
                    }
                    h = (h + 1) & (S - 1);
                }
            }
        }
        __syncthreads();
    }
    if (!NUMERIC) {
        if (local) atomicAdd(&counter, local);
        __syncthreads();
        if (tid == 0) row_nnz[row] = counter;
    } else {
        for (int k = tid; k < S; k += THREADS) {
            const int32_t key = keys[k];
            if (key != HASH_EMPTY) {
                const int pos = atomicAdd(&counter, 1);
                ccol[out0 + pos] = col_base(key);
                cval[out0 + pos] = vals[k];
            }
        }
    }
}

// ---- LDS hash kernel for SHORT rows of B: one wave per row of C, one 16-lane group per selected row of B ----------------------
// Counters of the flat-list kernel on the uniform configs[2] (round 4, profiles/r04_pmc_spgemm_uniform_lds.jsonl): 1 100
// instructions per row (540 VALU, 410 SALU, 125 LDS) for 256 products, s_waitcnt 8 % of the wave cycles, and the time did not
// move when the fetched bytes fell by a third (padded records): the kernel is bound by the INSTRUCTIONS of the flat walk (scan
// of the row lengths, position -> row search, stepping cursor, 64-bit positions), which pay off when rows of B are long or
// uneven.  When every row of B has at most 32 entries none of it is needed: lane group g of the wave takes row 4 i + g of the
// selected rows, lane l of the group its entry l (and l + 16) -- no scan, no search; four rows per group in flight.
template <typename T, int LOG2S, bool NUMERIC>
__global__ void __launch_bounds__(64)
    k_spgemm_grp(const int32_t* __restrict__ row_list, const int64_t* __restrict__ aptr,
                 const int64_t* __restrict__ ext0, const int32_t* __restrict__ extlen, const T* __restrict__ aval,
                 const int32_t* __restrict__ bcol, const T* __restrict__ bval, int upper,
                 int64_t* __restrict__ row_nnz, const int64_t* __restrict__ cptr, int32_t* __restrict__ ccol,
                 T* __restrict__ cval, ColOut col_base)
{
    constexpr int S = 1 << LOG2S, GW = 16, NG = 64 / GW, U = 4;
    __shared__ __attribute__((aligned(16))) int32_t keys[S];
    __shared__ __attribute__((aligned(16))) T vals[NUMERIC ? S : 1];
    __shared__ int64_t qlo[64];
    __shared__ int qlen[64];
    __shared__ T a_s[NUMERIC ? 64 : 1];
    const int lane = threadIdx.x, g = lane / GW, l16 = lane % GW;
    const int32_t row = row_list[blockIdx.x];
    {
        u32x4 e4 = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, z4 = {0u, 0u, 0u, 0u};
        if constexpr (S >= 256) {
            for (int k = lane; k < S / 4; k += 64) reinterpret_cast<u32x4*>(keys)[k] = e4;
            if (NUMERIC)
                for (int k = lane; k < (int)(S * sizeof(T) / 16); k += 64) reinterpret_cast<u32x4*>(vals)[k] = z4;
        } else {
            for (int k = lane; k < S; k += 64) {
                keys[k] = HASH_EMPTY;
                if (NUMERIC) vals[k] = vt<T>::zero();
            }
        }
    }
    int local = 0;
    const int64_t a0 = aptr[row], a1 = aptr[row + 1];
    int64_t out0 = 0;
    if (NUMERIC) out0 = cptr[row];
    for (int64_t base = a0; base < a1; base += 64) {
        wave_lds_sync();  // the previous chunk's readers are done (first chunk: the table is cleared)
        int len = 0;
        if (base + lane < a1) {
            qlo[lane] = ext0[base + lane];
            len = extlen[base + lane];
            if (NUMERIC) a_s[lane] = aval[base + lane];
        }
        qlen[lane] = len;
        const int n = (int)(a1 - base < 64 ? a1 - base : 64);
        int lmax = len;  // longest of the chunk's rows of B (the same in every lane)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_xor(lmax, d);
            lmax = o > lmax ? o : lmax;
        }
        wave_lds_sync();
        for (int s0 = 0; s0 < n; s0 += NG * U) {
            for (int off = 0; off < lmax; off += GW) {
                int32_t j[U];
                T v[U], av[U];
                int64_t q[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int sl = s0 + u * NG + g;  // < 64; slices at or beyond n have length 0
                    ok[u] = off + l16 < qlen[sl];
                    q[u] = ok[u] ? qlo[sl] + off + l16 : 0;
                    if (NUMERIC) av[u] = a_s[sl];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {  // unconditional, issued together
                    j[u] = bcol[q[u]];
                    if (NUMERIC) v[u] = bval[q[u]];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!ok[u]) j[u] = -1;
                    if (NUMERIC) v[u] = vt<T>::mul(av[u], v[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (j[u] < 0 || (upper && j[u] < row)) continue;
                    uint32_t h = hash_col(j[u], LOG2S);
                    for (;;) {
                        const int32_t old = atomicCAS(&keys[h], HASH_EMPTY, j[u]);
                        if (old == HASH_EMPTY || old == j[u]) {
                            if (NUMERIC) atomic_accum(&vals[h], v[u]);
                            else if (old == HASH_EMPTY) ++local;
                            break;
                        }
                        h = (h + 1) & (S - 1);
                    }
                }
            }
        }
    }
    if (!NUMERIC) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) local += __shfl_xor(local, d);
        if (lane == 0) row_nnz[row] = local;
    } else {
        wave_lds_sync();
        int written = 0;
        for (int k0 = 0; k0 < S; k0 += 64) {
            const int32_t key = keys[k0 + lane];
            int cnt;
            const int pos = wave_rank(key != HASH_EMPTY, cnt);
            if (key != HASH_EMPTY) {
                ccol[out0 + written + pos] = col_base(key);
                cval[out0 + written + pos] = vals[k0 + lane];
            }
            written += cnt;
        }
    }
}

// ---- one-pass kernel: no symbolic phase ----------------------------------------------------------------------------------
// When EVERY row of the product has few products (ub <= S / 2, S <= 1024) the symbolic pass is the same walk over B as the
// numeric one -- done twice only to learn where each row of C starts.  Here a wave forms its row ONCE (columns and values in
// its LDS table, as k_spgemm_lds) and the rows are placed by a DECOUPLED LOOK-BACK over the workgroups: a workgroup draws a
// ticket (its WAVES consecutive rows), publishes the number of entries of its rows as soon as they are known, adds up the
// published counts of the tickets before it until it meets one whose running total is already known, publishes its own
// running total and writes its rows at that position.  Tickets are drawn when a workgroup STARTS, so every ticket a
// workgroup waits for belongs to a workgroup that is running or done: no deadlock, whatever the dispatch order.  C's column
// and value arrays are allocated for the upper bound (sum of ub) before the kernel; nnz(C) is the last running total.
// Column order inside a row: table order (unspecified, as in the two-phase kernels).
constexpr unsigned long long LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_VALUE = (1ull << 62) - 1;

__device__ __forceinline__ int wave_min_int(int v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int n = __shfl_xor(v, d);
        v = n < v ? n : v;
    }
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
    return v;
}

template <typename T, int LOG2S, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    k_spgemm_onepass(int64_t rows, const int64_t* __restrict__ aptr, const int64_t* __restrict__ ub,
                     const int64_t* __restrict__ ext0, const int32_t* __restrict__ extlen, const T* __restrict__ aval,
                     const int32_t* __restrict__ bcol, const T* __restrict__ bval, int upper,
                     unsigned long long* __restrict__ ticket_counter, unsigned long long* __restrict__ flags,
                     int64_t* __restrict__ cptr, int32_t* __restrict__ ccol, T* __restrict__ cval)
{
    constexpr int S = 1 << LOG2S;
    __shared__ int32_t keys_all[WAVES][S];
    __shared__ T vals_all[WAVES][S];
    __shared__ int64_t qlo_all[WAVES][64];
    __shared__ T a_all[WAVES][64];
    __shared__ int inc_all[WAVES][64];
    __shared__ int row_n[WAVES];
    __shared__ long long block_excl, ticket_s;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int32_t* keys = keys_all[wave];
    T* vals = vals_all[wave];
    int64_t* qlo = qlo_all[wave];
    T* a_s = a_all[wave];
    int* inc = inc_all[wave];
    if (tid == 0) ticket_s = (long long)atomicAdd(ticket_counter, 1ull);
    __syncthreads();
    const int64_t blk = ticket_s;
    const int64_t row = blk * WAVES + wave;
    int nnz_row = 0, log2e = 6;
    if (row < rows) {  // whole wave
        // table sized for THIS row: the smallest power of two >= 2 ub (>= 64), so that a short row among long ones clears
        // and compacts what it needs
        const int64_t my_ub = ub[row];
        while (((int64_t)1 << log2e) < 2 * my_ub && log2e < LOG2S) ++log2e;
        const int se = 1 << log2e;
        for (int k = lane; k < se; k += 64) {
            keys[k] = HASH_EMPTY;
            vals[k] = vt<T>::zero();
        }
        int local = 0;
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        for (int64_t base = a0; base < a1; base += 64) {
            int len = 0;
            wave_lds_sync();  // the previous chunk's readers are done with qlo / a_s / inc (first chunk: the table is cleared)
            if (base + lane < a1) {
                qlo[lane] = ext0[base + lane];
                len = extlen[base + lane];
                a_s[lane] = aval[base + lane];
            }
            int v = len;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int n = __shfl_up(v, d);
                if (lane >= d) v += n;
            }
            inc[lane] = v;
            const int total = __shfl(v, 63);
            wave_lds_sync();
            for (int g0 = 0; g0 < total; g0 += 64 * LDS_UNROLL) {
                const int f0 = g0 + lane;
                FlatCursor<64> cur;
                int64_t qb = 0;
                T av = vt<T>::zero();
                if (f0 < total) {
                    cur.seek(inc, f0);
                    qb = qlo[cur.l];
                    av = a_s[cur.l];
                }
                int64_t q[LDS_UNROLL];
                T avs[LDS_UNROLL];
#pragma unroll
                for (int u = 0; u < LDS_UNROLL; ++u) {
                    const int f = f0 + u * 64;
                    q[u] = 0;
                    avs[u] = av;
                    if (f < total) {
                        if (u && cur.advance(inc, f)) {
                            qb = qlo[cur.l];
                            av = a_s[cur.l];
                        }
                        q[u] = qb + (f - cur.lo);
                        avs[u] = av;
                    }
                }
                int32_t j[LDS_UNROLL];
                T pv[LDS_UNROLL];
#pragma unroll
                for (int u = 0; u < LDS_UNROLL; ++u) {  // unconditional, issued together
                    j[u] = bcol[q[u]];
                    pv[u] = bval[q[u]];
                }
#pragma unroll
                for (int u = 0; u < LDS_UNROLL; ++u) {
                    if (f0 + u * 64 >= total) j[u] = -1;
                    pv[u] = vt<T>::mul(avs[u], pv[u]);
                }
#pragma unroll
                for (int u = 0; u < LDS_UNROLL; ++u) {
                    if (j[u] < 0 || (upper && j[u] < row)) continue;
                    uint32_t h = hash_col(j[u], log2e);
                    for (;;) {
                        const int32_t old = atomicCAS(&keys[h], HASH_EMPTY, j[u]);
                        if (old == HASH_EMPTY || old == j[u]) {
                            atomic_accum(&vals[h], pv[u]);
                            if (old == HASH_EMPTY) ++local;
                            break;
                        }
                        h = (h + 1) & (se - 1);
                    }
                }
            }
        }
        nnz_row = (int)wave_sum_i64(local);
    }
    if (lane == 0) row_n[wave] = nnz_row;
    __syncthreads();
    if (wave == 0) {
        long long agg = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) agg += row_n[w];
        long long excl = 0;
        if (blk == 0) {
            if (lane == 0) {
                agent_store(&flags[0], LB_PREFIX | (unsigned long long)agg);
                cptr[0] = 0;
            }
        } else {
            if (lane == 0) agent_store(&flags[blk], LB_AGG | (unsigned long long)agg);
            int64_t look = blk - 1;
            for (;;) {  // 64 earlier tickets per step, nearest first
                const int64_t idx = look - lane;
                const unsigned long long v = idx >= 0 ? agent_load(&flags[idx]) : LB_PREFIX;  // before ticket 0: total 0
                const int status = (int)(v >> 62);
                const int first_pfx = wave_min_int(status == 2 ? lane : 64);
                const int first_inv = wave_min_int(status == 0 ? lane : 64);
                const int limit = first_pfx < 64 ? first_pfx : 63;
                if (first_inv <= limit) {  // a count this step needs is not published yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += wave_sum_i64(lane <= limit ? (long long)(v & LB_VALUE) : 0ll);
                if (first_pfx < 64) break;
                look -= 64;
            }
            if (lane == 0) agent_store(&flags[blk], LB_PREFIX | (unsigned long long)(excl + agg));
        }
        if (lane == 0) block_excl = excl;
    }
    __syncthreads();
    if (row < rows) {
        int64_t out0 = block_excl;
        for (int w = 0; w < wave; ++w) out0 += row_n[w];
        const int se = 1 << log2e;
        int written = 0;
        for (int k0 = 0; k0 < se; k0 += 64) {
            const int32_t key = keys[k0 + lane];
            int cnt;
            const int pos = wave_rank(key != HASH_EMPTY, cnt);
            if (key != HASH_EMPTY) {
                ccol[out0 + written + pos] = key;
                cval[out0 + written + pos] = vals[k0 + lane];
            }
            written += cnt;
        }
        if (lane == 0) cptr[row + 1] = out0 + nnz_row;
    }
}

// ---- global-memory hash kernel: persistent workgroups, one slab each ------------------------------
template <typename T>
__device__ __forceinline__ T load_l2(const T* p)
{
    return __builtin_nontemporal_load(p);
}

// Atomics on a table that only ONE workgroup touches while it is live: workgroup scope is formally
// sufficient and lets the operation complete in the XCD's L2; the default (agent scope) atomics
// are performed at the memory side of the fabric so that all eight XCDs agree, which is an order
// of magnitude slower.  The table is cleared before / read back after with L1-bypassing accesses.
__device__ __forceinline__ int32_t cas_wg(int32_t* p, int32_t expected, int32_t desired)
{
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_WORKGROUP);
    return expected;
}
template <typename R>
__device__ __forceinline__ void add_wg(R* p, R x)
{
    __hip_atomic_fetch_add(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename T>
__device__ __forceinline__ void accum_wg(T* p, T x)
{
    add_wg(p, x);
}
template <typename R>
__device__ __forceinline__ void accum_wg(cx<R>* p, cx<R> x)
{
    add_wg(&p->re, x.re);
    add_wg(&p->im, x.im);
}

template <typename T, bool NUMERIC>
__global__ void __launch_bounds__(1024)
    k_spgemm_global(int64_t nbig, const int32_t* __restrict__ row_list, const int64_t* __restrict__ cnt,
                    int64_t ncols, const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol,
                    const T* __restrict__ aval, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol,
                    const T* __restrict__ bval, int gw, int upper, int32_t* slab_keys, T* slab_vals, int64_t slab,
                    int64_t* __restrict__ row_nnz, const int64_t* __restrict__ cptr, int32_t* __restrict__ ccol,
                    T* __restrict__ cval, unsigned long long* work_counter, ColOut col_base)
{
    __shared__ int counter;
    __shared__ long long next_idx;
    const int tid = threadIdx.x;
    const int threads = blockDim.x;
    int32_t* keys = slab_keys + (int64_t)blockIdx.x * slab;
    T* vals = NUMERIC ? slab_vals + (int64_t)blockIdx.x * slab : nullptr;
    for (;;) {
        // rows are listed largest class first; workgroups pull the next one when they are free
        __syncthreads();
        if (tid == 0) next_idx = (long long)atomicAdd(work_counter, 1ull);
        __syncthreads();
        const int64_t idx = next_idx;
        if (idx >= nbig) break;
        const int32_t row = row_list[idx];
        int64_t c = cnt[row];
        if (c > ncols) c = ncols;
        int log2s = 2;
        while (((int64_t)1 << log2s) < 2 * c) ++log2s;
        const int64_t S = (int64_t)1 << log2s;  // <= slab by construction
        for (int64_t k = tid; k < S; k += threads) {
            keys[k] = HASH_EMPTY;
            if (NUMERIC) vals[k] = vt<T>::zero();
        }
        if (tid == 0) counter = 0;
        __syncthreads();
        const int group = tid / gw, ngroups = threads / gw, gl = tid % gw;
        int local = 0;
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        for (int64_t p = a0 + group; p < a1; p += ngroups) {
            const int32_t kk = acol[p];
            T a = vt<T>::zero();
            if (NUMERIC) a = aval[p];
            const int64_t b0 = bptr[kk], b1 = bptr[kk + 1];
            for (int64_t q = b0 + gl; q < b1; q += gw) {
                const int32_t j = bcol[q];
                if (upper && j < row) continue;
                uint32_t h = hash_col(j, log2s);
                for (;;) {
                    const int32_t old = cas_wg(&keys[h], HASH_EMPTY, j);
                    if (old == HASH_EMPTY || old == j) {
                        if (NUMERIC) accum_wg(&vals[h], vt<T>::mul(a, bval[q]));
                        else if (old == HASH_EMPTY) ++local;
                        break;
                    }
                    h = (h + 1) & (uint32_t)(S - 1);
                }
            }
        }
        if (!NUMERIC) {
            if (local) atomicAdd(&counter, local);
            __syncthreads();
            if (tid == 0) row_nnz[row] = counter;
            __syncthreads();
        } else {
            __syncthreads();
            const int64_t base = cptr[row];
            for (int64_t k = tid; k < S; k += threads) {
                // the table was updated by L2 atomics: read it back past the (possibly stale) L1
                const int32_t key = load_l2(&keys[k]);
                if (key != HASH_EMPTY) {
                    const int pos = atomicAdd(&counter, 1);
                    ccol[base + pos] = col_base(key);
                    T v;
                    if (vt<T>::is_complex) {
                        using R = typename vt<T>::real;
                        const R* pr = reinterpret_cast<const R*>(&vals[k]);
                        R* vr = reinterpret_cast<R*>(&v);
                        vr[0] = load_l2(pr);
                        vr[1] = load_l2(pr + 1);
                    } else {
                        using R = typename vt<T>::real;
                        *reinterpret_cast<R*>(&v) = load_l2(reinterpret_cast<const R*>(&vals[k]));
                    }
                    cval[base + pos] = v;
                }
            }
            __syncthreads();
        }
    }
}

// Global-memory hash for the rows beyond the LDS bins, cooperative form: the tables of a whole batch
// of rows (one 2^log2s-slot table per row, cleared with memsets) live in a workspace, and the work
// is split by (row, slice of 64 A-nonzeros), so the hub rows of power-law matrices are spread over
// hundreds of workgroups instead of serialising on one.  L2 atomics make concurrent insertion into
// one table safe.  Symbolic: first insertions are counted per workgroup and added to row_nnz.
constexpr int GINS_THREADS = 256;
constexpr int GINS_ENTRIES = 64;  // A-nonzeros per workgroup

template <typename T, bool NUMERIC>
__global__ void __launch_bounds__(GINS_THREADS)
    k_spgemm_ginsert(const int32_t* __restrict__ row_list, const int64_t* __restrict__ aptr,
                     const int32_t* __restrict__ acol, const T* __restrict__ aval, const int64_t* __restrict__ bptr,
                     const int32_t* __restrict__ bcol, const T* __restrict__ bval, int gw, int upper, int32_t* keys_all,
                     T* vals_all, int log2s, unsigned long long* __restrict__ row_nnz,
                     const int64_t* __restrict__ item_off, int64_t nb)
{
    __shared__ int found;
    const int tid = threadIdx.x;
    // work item -> (row of the batch, slice of its A-nonzeros): item_off is the exclusive scan of the
    // rows' slice counts
    const int64_t item = blockIdx.x;
    int64_t lo = 0, hi = nb;  // largest t with item_off[t] <= item
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (item_off[mid] <= item) lo = mid; else hi = mid;
    }
    const int64_t t = lo;
    const int32_t row = row_list[t];
    const int64_t e0 = aptr[row] + (item - item_off[t]) * GINS_ENTRIES;
    int64_t e1 = aptr[row + 1];
    if (e0 + GINS_ENTRIES < e1) e1 = e0 + GINS_ENTRIES;
    if (!NUMERIC) {
        if (tid == 0) found = 0;
        __syncthreads();
    }
    const int64_t S = (int64_t)1 << log2s;
    int32_t* keys = keys_all + t * S;
    T* vals = NUMERIC ? vals_all + t * S : nullptr;
    const int group = tid / gw, ngroups = GINS_THREADS / gw, gl = tid % gw;
    int local = 0;
    for (int64_t p = e0 + group; p < e1; p += ngroups) {
        const int32_t kk = acol[p];
        T a = vt<T>::zero();
        if (NUMERIC) a = aval[p];
        const int64_t b0 = bptr[kk], b1 = bptr[kk + 1];
        for (int64_t q = b0 + gl; q < b1; q += gw) {
            const int32_t j = bcol[q];
            if (upper && j < row) continue;
            uint32_t h = hash_col(j, log2s);
            for (;;) {
                const int32_t old = atomicCAS(&keys[h], HASH_EMPTY, j);
                if (old == HASH_EMPTY || old == j) {
                    if (NUMERIC) atomic_accum(&vals[h], vt<T>::mul(a, bval[q]));
                    else if (old == HASH_EMPTY) ++local;
                    break;
                }
                h = (h + 1) & (uint32_t)(S - 1);
            }
        }
    }
    if (!NUMERIC) {
        if (local) atomicAdd(&found, local);
        __syncthreads();
        if (tid == 0 && found) atomicAdd(&row_nnz[row], (unsigned long long)found);
    }
}

__global__ void k_gins_items(const int32_t* __restrict__ row_list, const int64_t* __restrict__ aptr, int64_t nb,
                             int64_t* __restrict__ items)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const int32_t row = row_list[t];
    items[t] = (aptr[row + 1] - aptr[row] + GINS_ENTRIES - 1) / GINS_ENTRIES;
}

// compaction of a batch of tables into C: workgroup (slice of 4096 slots, row); positions inside a row
// come from a per-row cursor (one atomic per workgroup), so the column order is arbitrary
constexpr int GCOMP_SLOTS = 4096;
template <typename T>
__global__ void __launch_bounds__(256)
    k_spgemm_gcompact(const int32_t* __restrict__ row_list, const int32_t* keys_all, const T* vals_all, int log2s,
                      unsigned long long* __restrict__ cursor, const int64_t* __restrict__ cptr,
                      int32_t* __restrict__ ccol, T* __restrict__ cval, ColOut col_base)
{
    __shared__ int count;
    __shared__ long long base;
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.y;
    const int64_t S = (int64_t)1 << log2s;
    const int64_t k0 = (int64_t)blockIdx.x * GCOMP_SLOTS;
    if (k0 >= S) return;
    const int32_t* keys = keys_all + t * S;
    const T* vals = vals_all + t * S;
    if (tid == 0) count = 0;
    __syncthreads();
    constexpr int PER = GCOMP_SLOTS / 256;
    int32_t key[PER];
    int pos[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int64_t k = k0 + tid + (int64_t)u * 256;
        key[u] = (k < S) ? keys[k] : HASH_EMPTY;  // written by an earlier kernel: plain loads are safe
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) pos[u] = (key[u] != HASH_EMPTY) ? atomicAdd(&count, 1) : -1;
    __syncthreads();
    if (tid == 0 && count) base = (long long)atomicAdd(&cursor[t], (unsigned long long)count);
    __syncthreads();
    if (!count) return;
    const int32_t row = row_list[t];
    const int64_t out0 = cptr[row] + base;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        if (pos[u] >= 0) {
            const int64_t k = k0 + tid + (int64_t)u * 256;
            ccol[out0 + pos[u]] = col_base(key[u]);
            cval[out0 + pos[u]] = vals[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Big rows without global atomics (B with at most ~1.2 M columns, rows of B sorted for the numeric part).
//  * The set of columns of a row of C is an n-bit bitmap in LDS (n <= ~1.2 M fits the 160 KiB of a
//    CU): a product only ORs a bit.  Symbolic phase: nnz of the row = popcount of the bitmap.
//  * Numeric phase: the same bitmap is rebuilt (it costs a few ms for a whole power-law matrix) and cut
//    into P = ceil(nnz_i / CAP) column RANGES holding exactly CAP distinct columns each (the last one
//    the rest).  One workgroup per (row, range): it finds, by binary search in the sorted rows of B,
//    the slice of every B row that falls in its range, walks the slices as one flat list of products
//    (so hub rows of B spread over the whole workgroup), accumulates them in an LDS hash table that
//    cannot overflow (CAP = half its slots) and writes its CAP entries at cptr[row] + range * CAP --
//    no cursor, no global atomic, every product read once.
// ------------------------------------------------------------------------------------------------
#ifndef MI_PART_UNROLL
#define MI_PART_UNROLL 2
#endif
#ifndef MI_BITMAP_UNROLL
#define MI_BITMAP_UNROLL 16
#endif
#ifndef MI_PART_THREADS
#define MI_PART_THREADS 512
#endif
constexpr int PART_THREADS = MI_PART_THREADS;
#ifndef MI_PART_GROUP
#define MI_PART_GROUP 1
#endif
#ifndef MI_PART_XCD_RUN
#define MI_PART_XCD_RUN 32
#endif
constexpr int PART_XCD_RUN = MI_PART_XCD_RUN;  // consecutive work items per XCD (0: round-robin)
constexpr int PART_GROUP = MI_PART_GROUP;  // consecutive ranges of a big row per workgroup (k_spgemm_part)
constexpr int PART_UNROLL = MI_PART_UNROLL;
constexpr int BITMAP_UNROLL = MI_BITMAP_UNROLL;

// inc[0..N) non-decreasing (inclusive prefix sums), f < inc[N-1]: the number of entries <= f, i.e. the
// index of the slice that holds flat position f.  Fixed trip count, no divergence.
// word w of a row bitmap lives at bits[BITMAP_IDX(w)]: one pad word per 32, so that a thread walking its own
// run of consecutive words (rank computation below) and its neighbours hit different LDS banks
#define BITMAP_IDX(w) ((w) + ((w) >> 5))
static inline size_t bitmap_lds_bytes(int64_t ncols)
{
    const int64_t words = (ncols + 31) / 32;
    return sizeof(unsigned) * (size_t)(BITMAP_IDX(words) + 1);
}

// What k_spgemm_bitmap leaves behind besides the row length.  BM_COUNT: nothing.  BM_BOUNDS: the columns where every range of
// `cap` distinct columns starts (numeric phase: k_part_slices + k_spgemm_part).  BM_STORE (round 4): the bitmap itself, in global
// memory, and the number of set bits of every block of RANK_G columns (numeric phase: k_spgemm_rank).
enum { BM_COUNT = 0, BM_BOUNDS = 1, BM_STORE = 2 };
constexpr int RANK_G = 4096;           // columns per block: a (row, block) never holds more entries than the accumulators of k_spgemm_rank
constexpr int RANK_GW = RANK_G / 32;   // bitmap words per block
struct BitmapStore {
    unsigned* bm;        // nbig slots of wpr words: slot idx = position of the row in the big-row list of the symbolic phase
    int64_t wpr;         // words per slot: nblk * RANK_GW (>= the bitmap's words; the rest zero)
    uint16_t* blkcnt;    // nbig x nblk: set bits per block of RANK_G columns
    int32_t* slot_of;    // per row of A: its slot
    int nblk;
};

template <int MODE>
__global__ void __launch_bounds__(1024)
    k_spgemm_bitmap(int64_t nbig, const int32_t* __restrict__ row_list, int64_t ncols,
                    const int64_t* __restrict__ aptr, const int64_t* __restrict__ ext0,
                    const int32_t* __restrict__ extlen, const int32_t* __restrict__ bcol, int gw, int upper,
                    int64_t* __restrict__ row_nnz, const int64_t* __restrict__ boff, int64_t cap,
                    int32_t* __restrict__ bounds, int64_t* __restrict__ boff_by_row, unsigned long long* work_counter,
                    BitmapStore store)
{
    MI_DYN_SMEM(smem);
    unsigned* bits = reinterpret_cast<unsigned*>(smem);
    __shared__ int counter;
    __shared__ long long next_idx;
    __shared__ int scan[1024];
    __shared__ int wave_tot[16];
    __shared__ int64_t qlo[1024];
    const int tid = threadIdx.x, threads = 1024;  // launched with 1024 threads
    const int64_t words = (ncols + 31) / 32;
    const int64_t padded = BITMAP_IDX(words) + 1;
    // cap need not be a power of two (5/8 of the range kernel's table): n / cap as a multiplication, exact for n < 2^21 <= 2^40 / cap / 2^7
    const unsigned long long cap_magic = MODE == BM_BOUNDS ? (((1ull << 40) + (unsigned long long)cap - 1) / (unsigned long long)cap) : 0ull;
    for (;;) {
        __syncthreads();
        if (tid == 0) {
            next_idx = (long long)atomicAdd(work_counter, 1ull);
            counter = 0;
        }
        __syncthreads();
        const int64_t idx = next_idx;
        if (idx >= nbig) break;
        const int32_t row = row_list[idx];
        for (int64_t k = tid; k < padded; k += threads) bits[k] = 0u;
        __syncthreads();
        // the rows of B selected by this row of A, 1024 at a time, walked as one flat list of products (a hub
        // row of B spreads over the whole workgroup; BITMAP_UNROLL loads in flight per lane).  The extents of
        // the next 1024 rows are fetched while the current ones are walked.
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        int64_t b0_n = 0, b1_n = 0;
        auto fetch = [&](int64_t p) {
            b0_n = b1_n = 0;
            if (p < a1) {  // precomputed extents (k_row_ub): the lower triangle is already cut off
                b0_n = ext0[p];
                b1_n = b0_n + extlen[p];
            }
        };
        fetch(a0 + tid);
        for (int64_t base = a0; base < a1; base += 1024) {
            qlo[tid] = b0_n;
            const int len = (int)(b1_n - b0_n);
            fetch(base + 1024 + tid);
            block_scan_inclusive<1024>(len, scan, wave_tot, tid);
            const int* inc = scan;
            const int total = inc[1023];
            for (int g0 = 0; g0 < total; g0 += 1024 * BITMAP_UNROLL) {
                int32_t j[BITMAP_UNROLL];
                const int f0 = g0 + (tid >> 6) * 64 * BITMAP_UNROLL + (tid & 63);
                FlatCursor<1024> cur;
                int64_t qb = 0;
                if (f0 < total) {
                    cur.seek(inc, f0);
                    qb = qlo[cur.l];
                }
                int64_t q[BITMAP_UNROLL];
#pragma unroll
                for (int u = 0; u < BITMAP_UNROLL; ++u) {
                    const int f = f0 + u * 64;
                    q[u] = 0;
                    if (f < total) {
                        if (u && cur.advance(inc, f)) qb = qlo[cur.l];
                        q[u] = qb + (f - cur.lo);
                    }
                }
#pragma unroll
                for (int u = 0; u < BITMAP_UNROLL; ++u) j[u] = bcol[q[u]];  // unconditional, issued together
#pragma unroll
                for (int u = 0; u < BITMAP_UNROLL; ++u)
                    if (f0 + u * 64 >= total) j[u] = -1;
#pragma unroll
                for (int u = 0; u < BITMAP_UNROLL; ++u)
                    if (j[u] >= 0 && !(upper && j[u] < row)) atomicOr(&bits[BITMAP_IDX(j[u] >> 5)], 1u << (j[u] & 31));
            }
            __syncthreads();
        }
        if constexpr (MODE == BM_COUNT) {
            int local = 0;
            for (int64_t k = tid; k < padded; k += threads) local += __popc(bits[k]);  // pad words are zero
            if (local) atomicAdd(&counter, local);
            __syncthreads();
            if (tid == 0) row_nnz[row] = counter;
        } else if constexpr (MODE == BM_STORE) {
            // the bitmap goes out as it is (coalesced; 128 KiB for 2^20 columns) and so do the set bits per block of RANK_G columns:
            // the numeric phase cuts the row into runs of blocks, re-reads their bits and turns a column into its RANK in the run
            // -- the position of the entry in the (sorted) row of C -- with one LDS read and a popcount
            unsigned* dst = store.bm + idx * store.wpr;
            for (int64_t k = tid; k < store.wpr; k += threads) dst[k] = k < words ? bits[BITMAP_IDX(k)] : 0u;
            int wave_sum = 0;
            for (int b = tid >> 6; b < store.nblk; b += 16) {
                int c = 0;
                for (int i = tid & 63; i < RANK_GW; i += 64) {
                    const int64_t k = (int64_t)b * RANK_GW + i;
                    if (k < words) c += __popc(bits[BITMAP_IDX(k)]);
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
                if ((tid & 63) == 0) store.blkcnt[idx * store.nblk + b] = (uint16_t)c;  // <= RANK_G = 4096
                wave_sum += c;
            }
            if ((tid & 63) == 0 && wave_sum) atomicAdd(&counter, wave_sum);
            __syncthreads();
            if (tid == 0) {
                row_nnz[row] = counter;
                store.slot_of[row] = (int32_t)idx;
            }
        } else {
            // rank of every set bit -> the column where each range of `cap` distinct columns starts
            const int64_t per = (words + threads - 1) / threads;
            const int64_t w0 = (int64_t)tid * per, w1 = w0 + per < words ? w0 + per : words;
            int local = 0;
            for (int64_t k = w0; k < w1; ++k) local += __popc(bits[BITMAP_IDX(k)]);
            block_scan_inclusive<1024>(local, scan, wave_tot, tid);
            int64_t rank = scan[tid] - local;  // set bits before this thread's words
            int32_t* out = bounds + boff[idx];
            if (tid == 0) {
                out[0] = 0;
                row_nnz[row] = scan[1023];
                boff_by_row[row] = boff[idx];
            }
            for (int64_t k = w0; k < w1; ++k) {
                unsigned w = bits[BITMAP_IDX(k)];
                const int c = __popc(w);
                if (c) {
                    // boundaries b*cap (b >= 1) with rank <= b*cap < rank + c
                    int64_t b = (int64_t)(((unsigned long long)(rank + cap - 1) * cap_magic) >> 40);  // ceil(rank / cap), exact for rank < 2^21
                    if (b == 0) b = 1;
                    int taken = 0;  // set bits of the word already skipped
                    for (; b * cap < rank + c; ++b) {
                        const int want = (int)(b * cap - rank);  // 0-based index of the set bit inside the word
                        for (; taken < want; ++taken) w &= w - 1;
                        out[b] = (int32_t)(k * 32 + __builtin_ctz(w));
                    }
                }
                rank += c;
            }
        }
    }
}

__global__ void k_part_items(const int32_t* __restrict__ row_list, const int64_t* __restrict__ cnt, int64_t nb,
                             int64_t cap, int64_t clamp, int64_t* __restrict__ items)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    int64_t c = cnt[row_list[t]];
    if (c > clamp) c = clamp;
    items[t] = (c + cap - 1) / cap;
}

// Slice table for the numeric big-row kernel: for big row t with na nonzeros and P ranges,
//   bnd[slice_base[t] + p * na + e] = first position (absolute, in B's arrays) of B row acol[e] whose
// column is >= the start of range p (p = 0..P-1), and the end of that B row for p = P.
// One WAVE per (big row, block of 32 nonzeros of A, block of up to 64 range starts): the lanes hold the range
// starts and search the SAME sorted B row together, one nonzero after the other, so the row's cache lines are
// fetched once and every further probe is an L1 hit; the results go through an LDS tile so that the table is
// written along e (coalesced), the order k_spgemm_part reads it in.  (The first version gave every thread its
// own B row -- a bisection plus galloping per thread, 5+ private cache lines each: 54 ms and ~200 GB of fetches
// for the literal configs[2]; inside k_spgemm_part the same searches are dependent loads in a
// workgroup-synchronous phase, which is slower still.)
#ifndef MI_SLICE_EB
#define MI_SLICE_EB 16
#endif
constexpr int SLICE_EB = MI_SLICE_EB;  // nonzeros of A per wave
#ifndef MI_SLICE_ILP
#define MI_SLICE_ILP 2
#endif
constexpr int SLICE_ILP = MI_SLICE_ILP;  // bisections in flight per lane

__global__ void k_part_slice_sizes(const int32_t* __restrict__ row_list, const int64_t* __restrict__ item_off,
                                   const int64_t* __restrict__ aptr, int64_t nb, int64_t* __restrict__ n_work,
                                   int64_t* __restrict__ n_slice)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const int32_t row = row_list[t];
    const int64_t na = aptr[row + 1] - aptr[row];
    const int64_t P = item_off[t + 1] - item_off[t];
    n_work[t] = ((na + SLICE_EB - 1) / SLICE_EB) * ((P + 1 + 63) / 64);  // waves
    n_slice[t] = na * (P + 1);
}

__global__ void __launch_bounds__(256)
    k_part_slices(int64_t block_base, const int32_t* __restrict__ row_list, int64_t nb,
                  const int64_t* __restrict__ work_off,
                  const int64_t* __restrict__ item_off, const int32_t* __restrict__ bounds,
                  const int64_t* __restrict__ boff_by_row, const int64_t* __restrict__ aptr,
                  const int32_t* __restrict__ acol, const int64_t* __restrict__ bptr,
                  const int32_t* __restrict__ bcol, int upper, int32_t diag_shift, const int64_t* __restrict__ slice_base,
                  int32_t* __restrict__ bnd, const int32_t* __restrict__ work_t, const int32_t* __restrict__ cfloor)
{
    __shared__ int32_t tile_all[4][64][SLICE_EB + 1];  // [wave][range lane][nonzero]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int32_t (*tile)[SLICE_EB + 1] = tile_all[wave];
    const int64_t total = work_off[nb];
    const int64_t g = (block_base + blockIdx.x) * 4 + wave;  // wave index = work item
    if (g >= total) return;  // whole wave
    // row of this work item: precomputed (k_part_item_map) -- found here by a 19-step search over work_off it was ~10 us of
    // dependent loads at the start of every one of millions of waves (the same fix k_spgemm_part got in round 2)
    const int64_t t = work_t[g];
    const int32_t row = row_list[t];
    const int64_t a0 = aptr[row], na = aptr[row + 1] - a0;
    const int64_t P = item_off[t + 1] - item_off[t];
    const int64_t npb = (P + 1 + 63) / 64;
    const int64_t local = g - work_off[t];
    const int64_t eb = local / npb, pb = local - eb * npb;  // range block fastest: neighbouring waves share B rows
    const int64_t e0 = eb * SLICE_EB;
    const int ne = (int)(na - e0 < SLICE_EB ? na - e0 : SLICE_EB);
    const int64_t p0 = pb * 64;
    const int np = (int)(P + 1 - p0 < 64 ? P + 1 - p0 : 64);  // range starts p0 .. p0 + np - 1 (p == P: end of the row)
    // lanes = (nonzero sub-index, range): few ranges -> several nonzeros per step
    int pl_n = 1;
    while (pl_n < np) pl_n <<= 1;
    const int el_n = 64 / pl_n;
    const int pl = lane % pl_n, el = lane / pl_n;
    const int32_t* rb = bounds + boff_by_row[row];
    int32_t x = 0;
    const bool p_ok = pl < np;
    const bool is_end = p_ok && (p0 + pl == P);
    if (p_ok && !is_end) {
        x = rb[p0 + pl];
        if (upper && x < row - diag_shift) x = row - diag_shift;  // (diag_shift: B is a column panel with rebased columns)
        if (cfloor && x < cfloor[row]) x = cfloor[row];            // hub path: the columns left of the row's floor are not this path's
    }
    // SLICE_ILP nonzeros per lane at a time: their bisections advance in lockstep with unconditional loads, so
    // SLICE_ILP dependent chains are in flight per lane (one search after the other was ~10 dependent cache
    // round trips per nonzero, 32 nonzeros per wave in sequence: the kernel waited 85 % of its cycles)
    for (int eo = 0; eo < ne; eo += el_n * SLICE_ILP) {
        int64_t lo[SLICE_ILP], hi[SLICE_ILP];
#pragma unroll
        for (int u = 0; u < SLICE_ILP; ++u) {
            const int e = eo + u * el_n + el;
            lo[u] = hi[u] = 0;
            if (e < ne && p_ok) {
                const int32_t kk = acol[a0 + e0 + e];
                const int64_t b0 = bptr[kk], b1 = bptr[kk + 1];
                if (is_end) lo[u] = hi[u] = b1;
                else if (x <= 0) lo[u] = hi[u] = b0;
                else lo[u] = b0, hi[u] = b1;
            }
        }
        for (;;) {  // first position in [lo, hi) with column >= x, all searches together
            bool any = false;
            int32_t c[SLICE_ILP];
            int64_t mid[SLICE_ILP];
#pragma unroll
            for (int u = 0; u < SLICE_ILP; ++u) {
                mid[u] = (lo[u] + hi[u]) >> 1;
                c[u] = bcol[lo[u] < hi[u] ? mid[u] : 0];  // finished searches load a harmless element
            }
#pragma unroll
            for (int u = 0; u < SLICE_ILP; ++u) {
                if (lo[u] < hi[u]) {
                    if (c[u] < x) lo[u] = mid[u] + 1; else hi[u] = mid[u];
                    any = true;
                }
            }
            if (!any) break;
        }
#pragma unroll
        for (int u = 0; u < SLICE_ILP; ++u) {
            const int e = eo + u * el_n + el;
            if (e < ne && p_ok) tile[pl][e] = (int32_t)lo[u];
        }
    }
    // LDS is only shared inside the wave: lanes run in lockstep, a compiler / LDS fence is all that is needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // write out along e: lane -> (range sub-index, nonzero)
    int32_t* out = bnd + slice_base[t] + e0;
    const int wl = lane % SLICE_EB, wp = lane / SLICE_EB;  // 2 ranges x 32 nonzeros per step
    for (int po = 0; po < np; po += 64 / SLICE_EB) {
        const int pp = po + wp;
        // non-temporal: the table is read back by k_spgemm_part long after it has left every cache; plain stores displaced the
        // lines of B the bisections re-use (round 4, with the stores of k_spgemm_part: literal configs[2] 158.8 -> 152.5 ms)
        if (pp < np && wl < ne) __builtin_nontemporal_store(tile[pp][wl], &out[(p0 + pp) * na + wl]);
    }
}

// Everything a (row, range) workgroup needs about its row in ONE 64-byte load, and the row of every work
// item in one 4-byte load: the first version found its row by a 19-step binary search over the item offsets
// (dependent loads, ~10 us) at the start of each of millions of workgroups.
struct alignas(64) PartDesc {
    int64_t a0, na;         // the row's nonzeros in A
    int64_t item0;          // first work item of the row (range index = item - item0)
    int64_t slice_base;     // the row's slice table in bnd (k_part_slices)
    int64_t out0;           // cptr[row]
    int64_t boff;           // the row's range starts in bounds
    int32_t row, npass;
    int64_t group0;         // first workgroup of the row (a workgroup takes PART_GROUP consecutive ranges)
};

__global__ void k_part_desc(const int32_t* __restrict__ row_list, int64_t nb, const int64_t* __restrict__ item_off,
                            const int64_t* __restrict__ aptr, const int64_t* __restrict__ slice_base,
                            const int64_t* __restrict__ cptr, const int64_t* __restrict__ boff_by_row,
                            const int64_t* __restrict__ group_off, PartDesc* __restrict__ desc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const int32_t row = row_list[t];
    PartDesc d;
    d.a0 = aptr[row];
    d.na = aptr[row + 1] - d.a0;
    d.item0 = item_off[t];
    d.slice_base = slice_base[t];
    d.out0 = cptr[row];
    d.boff = boff_by_row[row];
    d.row = row;
    d.npass = (int32_t)(item_off[t + 1] - item_off[t]);
    d.group0 = group_off[t];
    desc[t] = d;
}

__global__ void k_part_item_map(int64_t n_items, int64_t nb, const int64_t* __restrict__ item_off,
                                int32_t* __restrict__ item_t)
{
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_items; g += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = nb;  // largest t with item_off[t] <= g
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (item_off[mid] <= g) lo = mid; else hi = mid;
        }
        item_t[g] = (int32_t)lo;
    }
}

// workgroups per big row: PART_GROUP consecutive ranges each
__global__ void k_part_groups(const int64_t* __restrict__ items, int64_t nb, int64_t* __restrict__ groups)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nb) groups[t] = (items[t] + PART_GROUP - 1) / PART_GROUP;
}

template <typename T, int LOG2S, bool PRE>
__global__ void __launch_bounds__(PART_THREADS)
    k_spgemm_part(int64_t item_base, int64_t n_items, const int32_t* __restrict__ item_t, const PartDesc* __restrict__ desc,
                  const int32_t* __restrict__ bounds, int64_t ncols, int64_t cap,
                  const int32_t* __restrict__ acol, const T* __restrict__ aval, const int64_t* __restrict__ bptr,
                  const int32_t* __restrict__ bcol, const T* __restrict__ bval, int upper,
                  const int32_t* __restrict__ bnd, int32_t* __restrict__ ccol, T* __restrict__ cval, ColOut col_base, int32_t diag_shift, const int32_t* __restrict__ cfloor)
{
    constexpr int S = 1 << LOG2S;
    constexpr int NT = PART_THREADS;
    __shared__ int32_t keys[S];
    __shared__ T vals[S];
    __shared__ int64_t qlo[NT];
    __shared__ T a_s[NT];
    __shared__ int pre[NT];
    __shared__ int wave_tot[NT / 64];
    __shared__ int n_out;
    const int tid = threadIdx.x;
    // XCD-affine order: workgroup b runs on XCD b % 8 (observed; speed only).  Runs of PART_XCD_RUN consecutive items --
    // the neighbouring ranges of one or two rows, which read neighbouring slices of the SAME rows of B and share the
    // cache lines at the slice boundaries (a slice is ~12 entries, a line 16-32) -- go to one XCD, back to back; the
    // runs go round-robin over the XCDs.  (One item per XCD in turn put neighbouring ranges on eight different L2s.)
    int64_t item;
    {
        const int64_t b = item_base + blockIdx.x;
        const int64_t local = b >> 3, blk = local / PART_XCD_RUN;
        item = PART_XCD_RUN > 0 ? (blk * 8 + (b & 7)) * PART_XCD_RUN + (local - blk * PART_XCD_RUN) : b;
        if (item >= n_items) return;  // the list is padded to a multiple of 8 runs
    }
    const PartDesc d = desc[item_t[item]];
    // PART_GROUP consecutive ranges of the row, one after the other: the descriptor is loaded once and the first
    // slices of the next range are fetched while the current one is walked.  Measured on the literal configs[2]
    // (profiles/r02_spgemm_group_ab.log): 1 / 2 / 4 / 8 / 16 ranges per workgroup = 237 / 246 / 247 / 246 / 249 ms
    // -- with four workgroups resident per CU the dependent chain item -> descriptor -> slices -> B entries is
    // already overlapped across workgroups -- so the default stays 1; kept as a tuning hook.
    const int64_t p_first = (item - d.group0) * PART_GROUP;
    const int64_t p_last = p_first + PART_GROUP < d.npass ? p_first + PART_GROUP : d.npass;
    const int32_t row = d.row;
    const int32_t* rb = bounds + d.boff;
    for (int k = tid; k < S; k += NT) {
        keys[k] = HASH_EMPTY;
        vals[k] = vt<T>::zero();
    }
    if (tid == 0) n_out = 0;
    __syncthreads();
    const int64_t a0 = d.a0, a1 = d.a0 + d.na;
    // slice of B row acol[p] that falls in the range's columns [c_lo, c_hi), and the A value: loaded one chunk ahead so
    // that the latency of these loads is hidden behind the product walk of the current chunk
    int64_t s_n = 0, e_n = 0;
    T a_n = vt<T>::zero();
    auto fetch = [&](int64_t pass, int64_t p) {
        s_n = e_n = 0;
        if (p < a1) {
            if constexpr (PRE) {
                const int32_t* sl0 = bnd + d.slice_base + pass * d.na;  // slices precomputed by k_part_slices
                s_n = sl0[p - a0];
                e_n = sl0[p - a0 + (a1 - a0)];
            } else {
                int32_t c_lo = rb[pass];
                const int64_t c_hi = pass + 1 < d.npass ? rb[pass + 1] : ncols;
                if (upper && c_lo < row - diag_shift) c_lo = row - diag_shift;
                if (cfloor && c_lo < cfloor[row]) c_lo = cfloor[row];
                const int32_t kk = acol[p];
                const int64_t b0 = bptr[kk], b1 = bptr[kk + 1];
                s_n = b0;
                e_n = b1;
                if (b0 < b1) {
                    if (c_lo > 0) s_n = lower_bound_col(bcol, b0, b1, c_lo);
                    if (c_hi < ncols) e_n = lower_bound_col(bcol, s_n, b1, (int32_t)c_hi);
                }
            }
            a_n = aval[p];
        }
    };
    fetch(p_first, a0 + tid);
    for (int64_t pass = p_first; pass < p_last; ++pass) {
        for (int64_t base = a0; base < a1; base += NT) {
            // 1. this chunk's slices into LDS, the next chunk's (or the next range's first) on their way
            qlo[tid] = s_n;
            a_s[tid] = a_n;
            const int len = (int)(e_n - s_n);
            if (base + NT < a1) fetch(pass, base + NT + tid);
            else if (pass + 1 < p_last) fetch(pass + 1, a0 + tid);
            // 2. inclusive scan of the slice lengths
            block_scan_inclusive<NT>(len, pre, wave_tot, tid);
            const int* inc = pre;
            const int total = inc[NT - 1];
            // 3. the slices as one flat list of products
            // (binary search per product, not the stepping cursor of the other kernels: the slices of ONE range are
            // short -- ~12 products -- and stepping over five of them per product costs more than the search)
            for (int f0 = tid; f0 < total; f0 += NT * PART_UNROLL) {
                // The loads of B are UNCONDITIONAL (positions past the end re-read the lane's first product and are
                // masked afterwards) and every use comes after the last load: with `if (f < total) { load; multiply }`
                // per product the compiler waited for product u before issuing the loads of product u + 1.
                int32_t j[PART_UNROLL];
                T v[PART_UNROLL];
                int64_t q[PART_UNROLL];
                T av[PART_UNROLL];
                int fs[PART_UNROLL], ls[PART_UNROLL];
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) fs[u] = f0 + u * NT < total ? f0 + u * NT : f0;
                flat_find_lockstep<NT, PART_UNROLL>(inc, fs, ls);
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) {
                    q[u] = qlo[ls[u]] + (fs[u] - (ls[u] ? inc[ls[u] - 1] : 0));
                    av[u] = a_s[ls[u]];
                }
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) {
                    j[u] = bcol[q[u]];
                    v[u] = bval[q[u]];
                }
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) {
                    if (f0 + u * NT >= total) j[u] = -1;
                    v[u] = vt<T>::mul(av[u], v[u]);
                }
                // first probe of every product issued back to back (the compare-and-swap returns a value: its latency
                // is paid once per batch, not once per product); the few collisions continue one at a time
                uint32_t hs[PART_UNROLL];
                int32_t was[PART_UNROLL];
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) {
                    hs[u] = hash_col(j[u] < 0 ? 0 : j[u], LOG2S);
                    was[u] = j[u];
                    if (j[u] >= 0) was[u] = atomicCAS(&keys[hs[u]], HASH_EMPTY, j[u]);
                }
#pragma unroll
                for (int u = 0; u < PART_UNROLL; ++u) {
                    if (j[u] < 0) continue;
                    uint32_t hsh = hs[u];
                    int32_t old = was[u];
                    while (!(old == HASH_EMPTY || old == j[u])) {
                        hsh = (hsh + 1) & (S - 1);
                        old = atomicCAS(&keys[hsh], HASH_EMPTY, j[u]);
                    }
                    atomic_accum(&vals[hsh], v[u]);
                }
            }
            __syncthreads();
        }
        // the range's entries out, the table left empty for the next range
        const int64_t out0 = d.out0 + pass * cap;
        for (int k = tid; k < S; k += NT) {
            const int32_t key = keys[k];
            if (key != HASH_EMPTY) {
                const int pos = atomicAdd(&n_out, 1);
                // non-temporal: 117 GB of C on the literal configs[2], written once -- as plain stores they evict the slices of
                // B that neighbouring ranges share (profiles/r04_spgemm_nt_stores_ab.log: 158.2 -> 153.5 ms)
                __builtin_nontemporal_store(col_base(key), &ccol[out0 + pos]);
                if constexpr (!vt<T>::is_complex) __builtin_nontemporal_store(vals[k], &cval[out0 + pos]);
                else cval[out0 + pos] = vals[k];
                keys[k] = HASH_EMPTY;
                vals[k] = vt<T>::zero();
            }
        }
        if (pass + 1 < p_last) {
            __syncthreads();
            if (tid == 0) n_out = 0;  // ordered before the next compaction by the barriers of the next range's walk
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Big rows, numeric phase, round 4: accumulate BY RANK (k_spgemm_rank).
//    The symbolic phase kept every big row's column bitmap (BM_STORE) and its set bits per block of RANK_G = 4096 columns.  A row
//    is cut into ITEMS = runs of consecutive blocks holding at most CAP entries of C and at most RANK_SPANB blocks.  One workgroup
//    per item:
//      * reads the item's piece of the bitmap (<= 16 KiB, coalesced), turns it into (word, set bits before it) records in LDS and
//        writes the item's column indices straight from the bits -- in increasing order, the whole row of C comes out SORTED;
//      * finds the slice of every selected row of B that falls into the item's columns by TABLE LOOKUP: items start and end at
//        block boundaries, and where block b starts inside every row of B (blkptr, k_blkptr) depends on B only -- no search per
//        (row of C, row of B, range) as k_part_slices does (2.4e9 bisections, 28 ms on the literal configs[2]);
//      * walks the slices as one flat product list and adds every product into acc[rank of its column] with ONE LDS atomic: the
//        rank is records[(j - c0) >> 5].before + popcount(word & bits below j) -- no hash probe, no key compare, no collision,
//        and the accumulators are all used (a hash table is half empty): CAP = 4096 entries of C per item instead of 1024,
//        a quarter of the items;
//      * writes acc[0 .. count) out in order (no compaction, no cursor atomics).
// ------------------------------------------------------------------------------------------------
#ifndef MI_RANK_THREADS
#define MI_RANK_THREADS 512
#endif
#ifndef MI_RANK_UNROLL
#define MI_RANK_UNROLL 2
#endif
#ifndef MI_RANK_SPANB
#define MI_RANK_SPANB 32
#endif
constexpr int RANK_THREADS = MI_RANK_THREADS;
constexpr int RANK_UNROLL = MI_RANK_UNROLL;
constexpr int RANK_SPANB = MI_RANK_SPANB;            // blocks per item at most (32: 131 072 columns, 4096 bitmap words)
constexpr int RANK_SPANW = RANK_SPANB * RANK_GW;
#ifndef MI_RANK_XCD_RUN
#define MI_RANK_XCD_RUN 4
#endif
constexpr int RANK_XCD_RUN = MI_RANK_XCD_RUN;  // consecutive groups per XCD
template <typename T>
constexpr int rank_cap() { return RANK_G; }  // accumulators per item (a block of RANK_G columns must fit): 32 KiB of fp64; complex double stays on k_spgemm_part

#ifndef MI_RANK_LONG
#define MI_RANK_LONG 64
#endif
#ifndef MI_RANK_SEG_UNROLL
#define MI_RANK_SEG_UNROLL 8
#endif
constexpr int RANK_LONG = MI_RANK_LONG;              // slices of at least this many entries are walked wave-wise, 64 entries at a time
constexpr int RANK_SEG_UNROLL = MI_RANK_SEG_UNROLL;  // segments in flight per wave

__device__ __forceinline__ int wave_uniform(int v)  // v is the same in every lane: keep it in a scalar register
{
    return __builtin_amdgcn_readfirstlane(v);
}

template <typename T>
__device__ __forceinline__ T wave_uniform_val(T v)  // the same for a value of any size that is a multiple of 4 bytes
{
    static_assert(sizeof(T) % 4 == 0, "wave_uniform_val: whole 32-bit words");
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) w[k] = __builtin_amdgcn_readfirstlane(w[k]);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// inclusive prefix sums of three ints per thread at once (one pair of barriers; see block_scan_inclusive)
template <int NT>
__device__ __forceinline__ void block_scan3(int& a, int& b, int& c, int (*wt)[NT / 64], int tid)
{
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int na = __shfl_up(a, d), nb = __shfl_up(b, d), nc = __shfl_up(c, d);
        if (lane >= d) {
            a += na;
            b += nb;
            c += nc;
        }
    }
    if (lane == 63) {
        wt[0][w] = a;
        wt[1][w] = b;
        wt[2][w] = c;
    }
    __syncthreads();
    int xa = 0, xb = 0, xc = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k)
        if (k < w) {
            xa += wt[0][k];
            xb += wt[1][k];
            xc += wt[2][k];
        }
    a += xa;
    b += xb;
    c += xc;
}

struct alignas(8) RankRec {
    unsigned bits, before;  // a word of the item's bitmap and the number of set bits in the words before it: one 8-byte LDS read per product
};
struct alignas(32) RankItem {
    int64_t out0;         // where the item's entries start in ccol / cval
    int32_t row, slot;    // row of A / C; the row's slot in the stored bitmaps
    int32_t b0, nb;       // blocks [b0, b0 + nb)
    int32_t count, pad;   // entries of C in the item
};

// blkptr[k * (nblk + 1) + b] = first position of row k of B (sorted) whose column is >= b * RANK_G; [nblk]: the end of the row.
// One wave per row; every entry that opens one or more blocks writes their starts.
__global__ void __launch_bounds__(256)
    k_blkptr(int64_t rows, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol, int nblk,
             int32_t* __restrict__ blkptr)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (row >= rows) return;
    const int64_t b0 = bptr[row], b1 = bptr[row + 1];
    int32_t* out = blkptr + row * (int64_t)(nblk + 1);
    if (b0 == b1) {
        for (int b = lane; b <= nblk; b += WAVE) out[b] = (int32_t)b0;
        return;
    }
    for (int64_t q = b0 + lane; q < b1; q += WAVE) {
        const int blk = bcol[q] / RANK_G;
        const int prev = q > b0 ? bcol[q - 1] / RANK_G : -1;
        for (int b = prev + 1; b <= blk; ++b) out[b] = (int32_t)q;
        if (q == b1 - 1)
            for (int b = blk + 1; b <= nblk; ++b) out[b] = (int32_t)b1;
    }
}

// Items of every big row: consecutive blocks are merged while they hold <= cap entries and span <= RANK_SPANB blocks; blocks
// without entries before the first / after the last entry of an item cost nothing and empty runs are no item at all.
// items == nullptr: count only (n_items[t]); else fill at items + item_off[t].
__global__ void __launch_bounds__(256)
    k_rank_items(int64_t nb, const int32_t* __restrict__ row_list, const int32_t* __restrict__ slot_of,
                 const uint16_t* __restrict__ blkcnt, int nblk, int cap, const int64_t* __restrict__ cptr,
                 const int64_t* __restrict__ item_off, int64_t* __restrict__ n_items, RankItem* __restrict__ items)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const int32_t row = row_list[t];
    const int32_t slot = slot_of[row];
    const uint16_t* cnt = blkcnt + (int64_t)slot * nblk;
    RankItem* out = items ? items + item_off[t] : nullptr;
    int64_t n = 0, done = 0;  // items so far, entries of C before the open item
    int cur_b0 = -1, cur = 0;
    auto emit = [&](int b_end) {
        if (out) {
            RankItem it;
            it.out0 = cptr[row] + done;
            it.row = row;
            it.slot = slot;
            it.b0 = cur_b0;
            it.nb = b_end - cur_b0;
            it.count = cur;
            it.pad = 0;
            out[n] = it;
        }
        ++n;
        done += cur;
        cur_b0 = -1;
        cur = 0;
    };
    for (int b = 0; b < nblk; ++b) {
        const int c = cnt[b];
        if (cur_b0 >= 0 && (cur + c > cap || b - cur_b0 >= RANK_SPANB)) emit(b);
        if (c == 0) continue;  // an empty block never opens an item (inside one it is skipped over for free)
        if (cur_b0 < 0) cur_b0 = b;
        cur += c;
    }
    if (cur_b0 >= 0) emit(nblk);
    if (!out) n_items[t] = n;
}

// One workgroup per GROUP = up to RANK_RUN consecutive items of one row, one after the other: the row's nonzeros of A (column,
// value, extent) are loaded ONCE and stay in registers (rows of up to NT nonzeros), and while item j is walked the loads item
// j + 1 starts with -- its piece of the bitmap and the two block starts per selected row of B -- are already in flight.  With one
// item per workgroup (first version) every item began with the chain  item -> row -> A's columns -> block starts -> entries of B,
// four dependent round trips with only two workgroups per CU (72 KiB of LDS each) to hide them behind: 22 us per item.
#ifndef MI_RANK_RUN
#define MI_RANK_RUN 16
#endif
constexpr int RANK_RUN = MI_RANK_RUN;
struct alignas(32) RankGroup {
    int64_t a0;        // the row's nonzeros in A
    int64_t item0;     // first item
    int32_t na, row;
    int32_t n, pad;    // items in the group
};

// groups of every big row (after k_rank_items has counted / filled the items)
__global__ void __launch_bounds__(256)
    k_rank_groups(int64_t nb, const int32_t* __restrict__ row_list, const int64_t* __restrict__ aptr,
                  const int64_t* __restrict__ item_off, const int64_t* __restrict__ group_off, int64_t* __restrict__ n_groups,
                  RankGroup* __restrict__ groups)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nb) return;
    const int64_t n = item_off[t + 1] - item_off[t];
    const int64_t ng = (n + RANK_RUN - 1) / RANK_RUN;
    if (!groups) {
        n_groups[t] = ng;
        return;
    }
    const int32_t row = row_list[t];
    for (int64_t g = 0; g < ng; ++g) {
        RankGroup r;
        r.a0 = aptr[row];
        r.na = (int32_t)(aptr[row + 1] - aptr[row]);
        r.row = row;
        r.item0 = item_off[t] + g * RANK_RUN;
        r.n = (int32_t)(n - g * RANK_RUN < RANK_RUN ? n - g * RANK_RUN : RANK_RUN);
        r.pad = 0;
        groups[group_off[t] + g] = r;
    }
}

template <typename T, int NT, int U>
__global__ void __launch_bounds__(NT, (sizeof(T) <= 8 ? 2 : 1) * NT / 256)  // two workgroups per CU (LDS): waves per SIMD
    k_spgemm_rank(int64_t group_base, int64_t n_groups, const RankGroup* __restrict__ groups, const RankItem* __restrict__ items,
                  const unsigned* __restrict__ bm, int64_t wpr, const int32_t* __restrict__ acol, const T* __restrict__ aval,
                  const int64_t* __restrict__ ext0, const int32_t* __restrict__ blkptr, int nblk1,
                  const int32_t* __restrict__ bcol, const T* __restrict__ bval, int32_t* __restrict__ ccol,
                  T* __restrict__ cval, ColOut col_base)
{
    constexpr int CAP = rank_cap<T>();
    constexpr int WPT = RANK_SPANW / NT;  // bitmap words per thread
    constexpr int OPT = CAP / NT;         // entries of C per thread
    constexpr int NW = NT / 64;
    static_assert(RANK_SPANW % NT == 0 && WPT >= 1 && CAP % NT == 0, "bitmap words and entries must divide over the threads");
    __shared__ RankRec rec[RANK_SPANW];
    __shared__ T acc[CAP];
    __shared__ int32_t qlo[NT];
    __shared__ int32_t qlen[NT];
    __shared__ T a_s[NT];
    __shared__ int inc_s[NT];       // short slices: inclusive prefix of their lengths (flat product list)
    __shared__ uint16_t lidx[NT];   // long slices, compacted: which slice, and the inclusive prefix of their 64-entry segments
    __shared__ int lseg[NT];
    __shared__ int wave_tot[NW];
    __shared__ int wave_tot3[3][NW];
    __shared__ int tot3[3];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = wave_uniform(tid >> 6);
    int64_t gi;  // XCD-affine runs of consecutive groups (see k_spgemm_part): the groups of a row share block starts and bitmap lines
    {
        const int64_t b = group_base + blockIdx.x;
        const int64_t local = b >> 3, blk = local / RANK_XCD_RUN;
        gi = RANK_XCD_RUN > 0 ? (blk * 8 + (b & 7)) * RANK_XCD_RUN + (local - blk * RANK_XCD_RUN) : b;
        if (gi >= n_groups) return;
    }
    const RankGroup g = groups[gi];
    const int64_t a0 = g.a0, a1 = g.a0 + g.na;
    // the first NT nonzeros of the row: column, extent start (upper triangle: already right of the diagonal), value -- kept
    const bool has0 = tid < g.na;
    int32_t k0 = 0, x0 = 0;
    T av0 = vt<T>::zero();
    if (has0) {
        k0 = acol[a0 + tid];
        x0 = (int32_t)ext0[a0 + tid];
        av0 = aval[a0 + tid];
    }
    // what an item starts with: its bits and, per nonzero of the first chunk, where its blocks start and end inside B's row
    auto load_bits = [&](const RankItem& it, unsigned (&w)[WPT]) {
        const int nw = it.nb * RANK_GW;
        const unsigned* src = bm + (int64_t)it.slot * wpr + (int64_t)it.b0 * RANK_GW + tid * WPT;
#pragma unroll
        for (int i = 0; i < WPT; ++i) w[i] = 0u;
        if (tid * WPT < nw) {  // nw is a multiple of RANK_GW = 128 words: a thread's WPT words are all inside or all outside
#pragma unroll
            for (int i = 0; i < WPT; ++i) w[i] = src[i];
        }
    };
    auto gather0 = [&](const RankItem& it, int32_t& s, int32_t& e) {
        s = e = 0;
        if (has0) {
            const int32_t* bp = blkptr + (int64_t)k0 * nblk1 + it.b0;
            s = bp[0];
            e = bp[it.nb];
        }
    };
    RankItem d = items[g.item0];
    unsigned w_n[WPT];
    int32_t s_g, e_g;
    load_bits(d, w_n);
    gather0(d, s_g, e_g);
    for (int jt = 0; jt < g.n; ++jt) {
        const int32_t c0 = d.b0 * RANK_G;
        // this item's loads are consumed, the next item's issued
        unsigned w[WPT];
        int local = 0;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            w[i] = w_n[i];
            local += __popc(w[i]);
        }
        int32_t s_n = s_g > x0 ? s_g : x0;
        int32_t e_n = e_g > s_n ? e_g : s_n;
        T a_n = av0;
        const RankItem d_next = items[g.item0 + (jt + 1 < g.n ? jt + 1 : jt)];
        if (jt + 1 < g.n) {
            load_bits(d_next, w_n);
            gather0(d_next, s_g, e_g);
        }
        // 1. bits -> rank records
        block_scan_inclusive<NT>(local, inc_s, wave_tot, tid);
        {
            int rank = inc_s[tid] - local;
#pragma unroll
            for (int i = 0; i < WPT; ++i) {
                rec[tid * WPT + i] = RankRec{w[i], (unsigned)rank};
                rank += __popc(w[i]);
            }
        }
        __syncthreads();
        // the item's column indices, entry k by thread k mod NT: the word that holds it by bisection over the records (all of a
        // thread's searches in lockstep), the bit inside the word by halving -- whole lines go out, in increasing order, so the
        // row of C is SORTED.  (Word by word -- every lane walking the set bits of its own words -- was 26 ms of divergent loops
        // on the literal configs[2], and its scattered stores another 6.)
        {
            int wi[OPT];
#pragma unroll
            for (int u = 0; u < OPT; ++u) wi[u] = 0;
#pragma unroll
            for (int step = RANK_SPANW / 2; step > 0; step >>= 1) {  // largest wi with rec[wi].before <= k
                unsigned probe[OPT];
#pragma unroll
                for (int u = 0; u < OPT; ++u) probe[u] = rec[wi[u] + step].before;
#pragma unroll
                for (int u = 0; u < OPT; ++u)
                    if ((int)probe[u] <= tid + u * NT) wi[u] += step;
            }
            RankRec r[OPT];
#pragma unroll
            for (int u = 0; u < OPT; ++u) r[u] = rec[wi[u]];
#pragma unroll
            for (int u = 0; u < OPT; ++u) {
                const int k = tid + u * NT;
                if (k >= d.count) continue;
                unsigned x = r[u].bits;
                int need = k - (int)r[u].before, pos = 0, c;  // the need-th (0-based) set bit of x
                c = __popc(x & 0xffffu);
                if (need >= c) { pos = 16; need -= c; }
                c = __popc((x >> pos) & 0xffu);
                if (need >= c) { pos += 8; need -= c; }
                c = __popc((x >> pos) & 0xfu);
                if (need >= c) { pos += 4; need -= c; }
                c = __popc((x >> pos) & 0x3u);
                if (need >= c) { pos += 2; need -= c; }
                c = (int)((x >> pos) & 1u);
                if (need >= c) pos += 1;
                ccol[d.out0 + k] = col_base(c0 + wi[u] * 32 + pos);
            }
        }
#pragma unroll
        for (int u = 0; u < OPT; ++u) acc[tid + u * NT] = vt<T>::zero();
        // 2. the selected rows of B, NT at a time.  A slice of >= RANK_LONG entries is cut into SEGMENTS of 64 and the segments
        //    are dealt to the waves in contiguous runs: the lanes of a wave sit on consecutive entries of ONE slice; slice, start
        //    and the value of A are wave-uniform (found once per segment, not once per product) and the loads are whole lines.
        //    The short slices are walked as one flat list of products (position -> slice by lockstep bisection).
        for (int64_t base = a0; base < a1; base += NT) {
            const int len = e_n - s_n;
            const bool is_long = len >= RANK_LONG;
            qlo[tid] = s_n;
            qlen[tid] = len;
            a_s[tid] = a_n;
            {  // the next chunk of a row of more than NT nonzeros (rare: hub rows of A): loaded while this one is walked
                const int64_t p = base + NT + tid;
                s_n = e_n = 0;
                if (p < a1) {
                    const int32_t* bp = blkptr + (int64_t)acol[p] * nblk1 + d.b0;
                    const int32_t xx = (int32_t)ext0[p];
                    const int32_t s = bp[0], e = bp[d.nb];
                    s_n = s > xx ? s : xx;
                    e_n = e > s_n ? e : s_n;
                    a_n = aval[p];
                }
            }
            int p_short = is_long ? 0 : len, p_seg = is_long ? (len + 63) >> 6 : 0, p_cnt = is_long ? 1 : 0;
            block_scan3<NT>(p_short, p_seg, p_cnt, wave_tot3, tid);
            inc_s[tid] = p_short;
            if (is_long) {
                lidx[p_cnt - 1] = (uint16_t)tid;
                lseg[p_cnt - 1] = p_seg;
            }
            if (tid == NT - 1) {
                tot3[0] = p_short;
                tot3[1] = p_seg;
                tot3[2] = p_cnt;
            }
            __syncthreads();
            const int total = tot3[0], tot_seg = tot3[1], n_long = tot3[2];
            // 2a + 2b in ONE loop: a round computes the addresses of up to RANK_SEG_UNROLL segments (long slices) and U flat positions
            // (short slices), issues all their loads together and then consumes them -- one memory round trip per round.  (As two
            // loops, one after the other, a chunk paid two: ~3 us each with two workgroups per CU to hide them behind.)
            int sg = 0, sg1 = 0, c = 0, c_end = 0, c_base = 0;
            if (tot_seg > 0) {
                const int per = (tot_seg + NW - 1) / NW;
                sg = wave * per;
                sg1 = sg + per < tot_seg ? sg + per : tot_seg;
                if (sg < sg1) {
                    int lo = 0, hi = n_long;  // first long slice whose segments end beyond sg
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (lseg[mid] <= sg) lo = mid + 1; else hi = mid;
                    }
                    // everything about a segment but the lane is the SAME in all lanes: kept in scalar registers
                    c = wave_uniform(lo);
                    c_end = wave_uniform(lseg[c]);
                    c_base = wave_uniform(c ? lseg[c - 1] : 0);
                }
            }
            const int* inc = inc_s;
            for (int g0 = 0; g0 < total || sg < sg1; g0 += NT * U, sg += RANK_SEG_UNROLL) {
                // addresses: segments
                int qb[RANK_SEG_UNROLL], nn[RANK_SEG_UNROLL];
                T sav[RANK_SEG_UNROLL];
#pragma unroll
                for (int u = 0; u < RANK_SEG_UNROLL; ++u) {
                    const int gg = sg + u;
                    qb[u] = 0;
                    nn[u] = 0;
                    sav[u] = vt<T>::zero();
                    if (gg < sg1) {
                        while (gg >= c_end) {
                            c_base = c_end;
                            ++c;
                            c_end = wave_uniform(lseg[c]);
                        }
                        const int sl = wave_uniform((int)lidx[c]);
                        const int off = (gg - c_base) << 6;
                        qb[u] = wave_uniform(qlo[sl]) + off;
                        nn[u] = wave_uniform(qlen[sl]) - off;
                        sav[u] = wave_uniform_val(a_s[sl]);
                    }
                }
                // addresses: flat positions
                const int f0 = g0 + tid;
                int32_t q[U];
                T av[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    q[u] = 0;
                    av[u] = vt<T>::zero();
                }
                if (g0 < total) {
                    int fs[U], ls[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) fs[u] = f0 + u * NT < total ? f0 + u * NT : (f0 < total ? f0 : 0);
                    flat_find_lockstep<NT, U>(inc, fs, ls);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        q[u] = qlo[ls[u]] + (fs[u] - (ls[u] ? inc[ls[u] - 1] : 0));
                        av[u] = a_s[ls[u]];
                    }
                }
                // loads, all together
                int32_t sj[RANK_SEG_UNROLL], j[U];
                T sv[RANK_SEG_UNROLL], v[U];
#pragma unroll
                for (int u = 0; u < RANK_SEG_UNROLL; ++u) {
                    const int32_t qq = lane < nn[u] ? qb[u] + lane : 0;
                    sj[u] = bcol[qq];
                    sv[u] = bval[qq];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    j[u] = bcol[q[u]];
                    v[u] = bval[q[u]];
                }
                // consume
#pragma unroll
                for (int u = 0; u < RANK_SEG_UNROLL; ++u) {
                    if (lane >= nn[u]) continue;
                    const int rel = sj[u] - c0;
                    const RankRec r = rec[rel >> 5];
                    const int rank = (int)r.before + __popc(r.bits & ((1u << (rel & 31)) - 1u));
                    atomic_accum(&acc[rank], vt<T>::mul(sav[u], sv[u]));
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (f0 + u * NT >= total) continue;
                    const int rel = j[u] - c0;
                    const RankRec r = rec[rel >> 5];
                    const int rank = (int)r.before + __popc(r.bits & ((1u << (rel & 31)) - 1u));
                    atomic_accum(&acc[rank], vt<T>::mul(av[u], v[u]));
                }
            }
            __syncthreads();
        }
        // 3. the values, in column order
#pragma unroll
        for (int u = 0; u < OPT; ++u)
            if (tid + u * NT < d.count) cval[d.out0 + tid + u * NT] = acc[tid + u * NT];
        d = d_next;
        __syncthreads();  // rec / acc are rewritten by the next item
    }
}

// ---- dense-output variant (spmmd): one wave per row, products scattered with L2 atomics ------------
template <typename T>
__global__ void __launch_bounds__(256)
    k_spmmd(int64_t rows, const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol,
            const T* __restrict__ aval, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol,
            const T* __restrict__ bval, T* __restrict__ C, int64_t c_rs, int64_t c_cs)
{
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (row >= rows) return;
    T* crow = C + row * c_rs;
    for (int64_t p = aptr[row]; p < aptr[row + 1]; ++p) {
        const int32_t kk = acol[p];
        const T a = aval[p];
        for (int64_t q = bptr[kk] + lane; q < bptr[kk + 1]; q += WAVE)
            atomic_accum(crow + (int64_t)bcol[q] * c_cs, vt<T>::mul(a, bval[q]));
    }
}

// Option "deterministic": the values of C recomputed in a FIXED order.  One wave owns a row of C (its columns already sorted):
// it takes the nonzeros of A's row one after the other, the lanes span the entries of the selected row of B, every product finds
// its place in the row by a search and is added with a global atomic -- atomics of one wave to one address are performed in the
// order they were issued (one queue to one L2 channel), so every entry is the same sequence of additions on every run
// (k_spmmd sums the same way).  Slow on hub rows (one wave walks all of the row's products); a validation mode.
template <typename T>
__global__ void __launch_bounds__(256)
    k_spgemm_values_det(int64_t row0, int64_t rows, const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol,
                        const T* __restrict__ aval, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol,
                        const T* __restrict__ bval, int upper, const int64_t* __restrict__ cptr,
                        const int32_t* __restrict__ ccol, T* __restrict__ cval)
{
    const int64_t row = row0 + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (row >= rows) return;
    const int64_t c0 = cptr[row], c1 = cptr[row + 1];
    for (int64_t p = aptr[row]; p < aptr[row + 1]; ++p) {
        const int32_t kk = acol[p];
        const T a = aval[p];
        for (int64_t q = bptr[kk] + lane; q < bptr[kk + 1]; q += WAVE) {
            const int32_t j = bcol[q];
            if (upper && j < row) continue;
            const int64_t pos = lower_bound_col(ccol, c0, c1, j);
            atomic_accum(cval + pos, vt<T>::mul(a, bval[q]));
        }
    }
}

template <typename T>
__global__ void k_fill_dense(T* C, int64_t r, int64_t cdim, int64_t c_rs, int64_t c_cs, T v)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= r * cdim) return;
    // consecutive threads -> consecutive memory
    const bool row_major = (c_cs == 1);
    const int64_t i = row_major ? t / cdim : t % r;
    const int64_t j = row_major ? t % cdim : t / r;
    C[i * c_rs + j * c_cs] = v;
}

#include "spgemm_hub.inc"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct Bins {
    int64_t n[NBINS] = {};
    int32_t* list[NBINS] = {};
};

// sum, maximum and the size-class histogram (bin_of) of `n` row counts in ONE pass and ONE copy back: out[0] += sum,
// out[1] = max, out[2 + k] += rows of class k (rows with count 0 are in no class, as in k_bin_rows)
__global__ void __launch_bounds__(256) k_row_stats(const int64_t* in, int64_t n, int64_t* out)
{
    __shared__ int64_t red_s[256], red_m[256];
    __shared__ int hist[NBINS];
    if (threadIdx.x < NBINS) hist[threadIdx.x] = 0;
    __syncthreads();
    int64_t s = 0, m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = in[i];
        s += v;
        if (v > m) m = v;
        if (v > 0) atomicAdd(&hist[bin_of(v)], 1);
    }
    red_s[threadIdx.x] = s;
    red_m[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red_s[threadIdx.x] += red_s[threadIdx.x + off];
            if (red_m[threadIdx.x + off] > red_m[threadIdx.x]) red_m[threadIdx.x] = red_m[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&out[0], (unsigned long long)red_s[0]);
        atomicMax((long long*)&out[1], (long long)red_m[0]);
    }
    if (threadIdx.x < NBINS && hist[threadIdx.x] > 0)
        atomicAdd((unsigned long long*)&out[2 + threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

struct RowStats {
    int64_t sum = 0, max = 0;
    int64_t bins[NBINS] = {};
};

static RowStats device_row_stats(const int64_t* in, int64_t n)
{
    Context& c = ctx();
    int64_t* d = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (2 + NBINS)));
    MI_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(int64_t) * (2 + NBINS), c.stream));
    if (n > 0) {
        const int64_t blocks = ceil_div(n, 1024) < 2048 ? ceil_div(n, 1024) : 2048;
        MI_LAUNCH(k_row_stats, dim3((unsigned)blocks), dim3(256), c.stream, in, n, d);
    }
    int64_t h[2 + NBINS] = {};
    MI_HIP_CHECK(hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    MI_HIP_CHECK(hipStreamSynchronize(c.stream));
    RowStats r;
    r.sum = h[0];
    r.max = h[1];
    for (int k = 0; k < NBINS; ++k) r.bins[k] = h[2 + k];
    return r;
}

// row lists per size class; the class sizes come from device_row_stats (no counting pass, no round trip of its own)
static Bins make_bins(const int64_t* cnt, int64_t rows, const RowStats& rs)
{
    Context& c = ctx();
    Bins b;
    int64_t* dcounts = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * NBINS));
    const dim3 grid((unsigned)ceil_div(rows > 0 ? rows : 1, 256));
    BinLists hl;
    for (int k = 0; k < NBINS; ++k) {
        b.n[k] = rs.bins[k];
        b.list[k] = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(b.n[k] + 1)));
        hl.l[k] = b.list[k];
    }
    MI_HIP_CHECK(hipMemsetAsync(dcounts, 0, sizeof(int64_t) * NBINS, c.stream));
    MI_LAUNCH(k_bin_rows, grid, dim3(256), c.stream, rows, cnt, dcounts, hl);  // the pointers travel as a kernel argument
    return b;
}

static int pick_gw(const Csr& B)
{
    const double avg = B.rows > 0 ? (double)B.nnz / (double)B.rows : 1.0;
    int gw = 4;
    while (gw < 64 && gw < avg) gw <<= 1;
    return gw;
}

__global__ void k_max_i64(const int64_t* in, int64_t n, int64_t* out)
{
    __shared__ int64_t red[256];
    int64_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (in[i] > m) m = in[i];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off && red[threadIdx.x + off] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicMax((long long*)out, (long long)red[0]);
}

static int64_t device_max(const int64_t* in, int64_t n)
{
    Context& c = ctx();
    int64_t* d = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t)));
    MI_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(int64_t), c.stream));
    if (n > 0) {
        const int64_t blocks = ceil_div(n, 256) < 1024 ? ceil_div(n, 256) : 1024;
        MI_LAUNCH(k_max_i64, dim3((unsigned)blocks), dim3(256), c.stream, in, n, d);
    }
    int64_t h = 0;
    MI_HIP_CHECK(hipMemcpyAsync(&h, d, sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    MI_HIP_CHECK(hipStreamSynchronize(c.stream));
    return h;
}

__global__ void k_row_len(const int64_t* ptr, int64_t rows, int64_t* len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) len[i] = ptr[i + 1] - ptr[i];
}

static int64_t device_max_row_len(const Csr& A)
{
    if (A.rows == 0) return 0;
    Context& c = ctx();
    int64_t* len = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)A.rows));
    MI_LAUNCH(k_row_len, dim3((unsigned)ceil_div(A.rows, 256)), dim3(256), c.stream, (const int64_t*)A.ptr, A.rows, len);
    return device_max(len, A.rows);
}

// HIP launches at most 2^32 - 1 threads per grid (more are silently not run): split a 1-D grid of `nblocks`
// workgroups of `threads` threads into launches of at most 2^31 threads; f(first block, number of blocks).
template <typename F>
static void launch_batched(int64_t nblocks, int threads, F&& f)
{
    const int64_t maxb = ((int64_t)1 << 31) / threads;
    for (int64_t off = 0; off < nblocks; off += maxb) f(off, nblocks - off < maxb ? nblocks - off : maxb);
}

// What the symbolic phase leaves for the numeric phase about the big rows (LDS bitmap path).
struct BigRows {
    int log2s = 11;            // table size of the numeric range kernel; a range holds cap = 2^log2s * MI_PART_FILL_8THS / 8 columns (5/8: not a power of two)
    int64_t cap = 1024;
    bool b_sorted = false;     // rows of B sorted (needed to cut B rows into column ranges by search)
    bool have_bounds = false;  // symbolic phase ran the bitmap kernel and stored the range starts
    DevBuf boff_by_row;        // int64[A.rows]: offset of a big row's range starts in `bounds`
    DevBuf bounds;             // int32: range starts, P_max = ceil(min(ub, cols) / cap) slots per big row
    DevBuf ext0, extlen;       // per nonzero of A: first counted entry of B's row (int64) and their number (int32) -- k_row_ub
    bool grp = false;          // every row of B has <= 32 entries: the LDS bins up to 512 products run k_spgemm_grp
    ColOut col_base;           // numeric phase: added to every column index written (B is a column panel of a wider matrix, spgemm_panels)
    const int32_t* cfloor = nullptr;  // hub path: per row of A, the first (relabelled) column the range path owns -- the dense blocks left of it are accumulated by k_hub_num
    // upper-triangle product of a column panel: the extents (ext0 / extlen) are cut at the diagonal shifted by the panel's first
    // column and the kernels that read them run as a full product; the two kernels that cut rows of B into column ranges from
    // B's own row pointer (k_part_slices, k_spgemm_part) clamp at row - diag_shift themselves
    bool panel_upper = false;
    int32_t diag_shift = 0;
    // round 4, accumulate-by-rank path (k_spgemm_rank): the symbolic phase's bitmaps and what the numeric phase needs with them
    bool have_rank = false;
    bool want_rank = false;    // the caller will order the result (mi_sparse_spmm_ordered): take the rank path when its bitmaps are affordable
    int64_t out_range_cap = 0;  // numeric phase, range-partitioned hash: the big rows of C were written as consecutive column ranges of this many entries
    int nblk = 0;              // blocks of RANK_G columns
    int64_t wpr = 0;           // bitmap words per stored row
    DevBuf bm_store;           // unsigned[nbig * wpr]
    DevBuf blkcnt;             // uint16[nbig * nblk]
    DevBuf slot_of;            // int32[A.rows]: slot of a big row in bm_store / blkcnt
    DevBuf blkptr;             // int32[B.rows * (nblk + 1)]: where every block starts inside every row of B (k_blkptr)
};

// `lists`: in -- row lists to use instead of binning `cnt` (the symbolic phase's, when every row sits in an LDS class: a table
// sized for the upper bound of a row holds its exact length too); out (symbolic phase) -- the lists it built
template <typename T, bool NUMERIC>
static void run_phase(const Csr& A, const Csr& B, int upper, const int64_t* cnt, const RowStats& rs, int64_t* row_nnz,
                      const int64_t* cptr, int32_t* ccol, T* cval, BigRows& big, Bins* lists = nullptr)
{
    Context& c = ctx();
    const int gw = pick_gw(B);
    const int gw64 = gw > 64 ? 64 : gw;
    const int64_t max_cnt = rs.max;
    Bins b;
    if (NUMERIC && lists) b = *lists;
    else {
        b = make_bins(cnt, A.rows, rs);
        if (lists) *lists = b;
    }
    const bool force_global = options().spgemm_force_global != 0;
    // big rows (beyond the numeric LDS tables): LDS bitmap in the symbolic phase, range-partitioned LDS hash in
    // the numeric one -- when B is narrow enough for a bitmap (and, numeric, its rows are sorted)
    const int first_big = sizeof(T) >= 16 ? 7 : 8;
    const bool use_bitmap = !force_global && options().spgemm_lds_parts &&
                            (NUMERIC ? (big.have_bounds || big.have_rank)
                                     : (bitmap_lds_bytes(B.cols) <= (size_t)140 * 1024 && B.nnz < ((int64_t)1 << 31)));
#define MI_SPGEMM_ARGS(list)                                                                                       \
    (const int32_t*)list, (const int64_t*)A.ptr, (const int32_t*)A.col, (const T*)A.val, (const int64_t*)B.ptr,   \
        (const int32_t*)B.col, (const T*)B.val
#define MI_SPGEMM_LDS_ARGS(list)                                                                                   \
    (const int32_t*)list, (const int64_t*)A.ptr, (const int64_t*)big.ext0.as<int64_t>(),                           \
        (const int32_t*)big.extlen.as<int32_t>(), (const T*)A.val, (const int32_t*)B.col, (const T*)B.val
    if (!force_global) {
#define MI_SPGEMM_BIN(k, LOG2S, THREADS, GW)                                                                       \
    if (b.n[k]) {                                                                                                  \
        launch_batched(b.n[k], THREADS, [&](int64_t off, int64_t nb) {                                             \
            MI_LAUNCH((k_spgemm_lds<T, LOG2S, THREADS, NUMERIC>), dim3((unsigned)nb), dim3(THREADS), c.stream,       \
                      MI_SPGEMM_LDS_ARGS(b.list[k] + off), GW, (int)upper, row_nnz, cptr, ccol, cval, big.col_base); \
        });                                                                                                        \
        b.n[k] = 0;                                                                                                \
    }
#define MI_SPGEMM_GRP(k, LOG2S)                                                                                     \
    if (big.grp && b.n[k]) {                                                                                       \
        launch_batched(b.n[k], 64, [&](int64_t off, int64_t nb) {                                                  \
            MI_LAUNCH((k_spgemm_grp<T, LOG2S, NUMERIC>), dim3((unsigned)nb), dim3(64), c.stream,                     \
                      MI_SPGEMM_LDS_ARGS(b.list[k] + off), (int)upper, row_nnz, cptr, ccol, cval, big.col_base);     \
        });                                                                                                        \
        b.n[k] = 0;                                                                                                \
    }
        MI_SPGEMM_GRP(0, 6)
        MI_SPGEMM_GRP(1, 7)
        MI_SPGEMM_GRP(2, 8)
        MI_SPGEMM_GRP(3, 9)
        MI_SPGEMM_GRP(4, 10)
#undef MI_SPGEMM_GRP
        MI_SPGEMM_BIN(0, 6, 64, gw64)
        MI_SPGEMM_BIN(1, 7, 64, gw64)
        MI_SPGEMM_BIN(2, 8, 64, gw64)
        MI_SPGEMM_BIN(3, 9, MI_BIN3_THREADS, gw64)
        MI_SPGEMM_BIN(4, 10, MI_BIN4_THREADS, gw64)
        MI_SPGEMM_BIN(5, 11, 256, gw)
        MI_SPGEMM_BIN(6, 12, 256, gw)
        // bin 7 (<= 4096): 8192-slot table -- 32 KiB of keys + up to 64 KiB of values.  (Symbolic phase with
        // the bitmap available: every row the numeric phase will treat as big must pass through the bitmap.)
        if constexpr (!NUMERIC) {
            if (!(use_bitmap && first_big <= 7)) {
                MI_SPGEMM_BIN(7, 13, 1024, gw)
            }
            // bin 8 (<= 8192): only the keys fit: 16384-slot table = 64 KiB
            if (!use_bitmap) {
                MI_SPGEMM_BIN(8, 14, 1024, gw)
            }
        } else if constexpr (sizeof(T) <= 8) {
            MI_SPGEMM_BIN(7, 13, 1024, gw)
        }
#undef MI_SPGEMM_BIN
    }
    // Big rows through LDS (column bitmap + range-partitioned hash, see k_spgemm_bitmap / k_spgemm_part);
    // when B is too wide for a bitmap or its rows are not sorted they stay for the global-memory hash.
    if (use_bitmap) {
        int64_t nbig = 0;
        for (int k = first_big; k < NBINS; ++k) nbig += b.n[k];
        if (nbig) {
            int32_t* big_list = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)nbig));
            int64_t off = 0;
            for (int k = NBINS - 1; k >= first_big; --k) {  // largest class first
                if (!b.n[k]) continue;
                MI_HIP_CHECK(hipMemcpyAsync(big_list + off, b.list[k], sizeof(int32_t) * (size_t)b.n[k],
                                            hipMemcpyDeviceToDevice, c.stream));
                off += b.n[k];
                b.n[k] = 0;
            }
            int64_t* items = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
            int64_t* item_off = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
            if constexpr (!NUMERIC) {
                unsigned long long* counter = static_cast<unsigned long long*>(c.scratch_alloc(sizeof(unsigned long long)));
                MI_HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), c.stream));
                const int64_t nblocks = nbig < 512 ? nbig : 512;
                // range starts are recorded only if the numeric phase can use them (sorted B, sane size)
                int64_t n_bounds = 0;
                if (big.b_sorted) {
                    MI_LAUNCH(k_part_items, dim3((unsigned)ceil_div(nbig, 256)), dim3(256), c.stream,
                              (const int32_t*)big_list, cnt, nbig, big.cap, B.cols, items);
                    n_bounds = exclusive_scan_i64(items, item_off, nbig);
                }
                // round 4: keep the bitmaps for k_spgemm_rank when they (and the block starts of B's rows) are affordable
                bool rank_path = false;
                if (big.b_sorted && (options().spgemm_rank || big.want_rank) && sizeof(T) <= 8 && !big.cfloor) {
                    const int nblk = (int)ceil_div(B.cols, (int64_t)RANK_G);
                    const int64_t wpr = (int64_t)nblk * RANK_GW;
                    const size_t need = sizeof(unsigned) * (size_t)wpr * (size_t)nbig + sizeof(uint16_t) * (size_t)nblk * (size_t)nbig +
                                        sizeof(int32_t) * (size_t)(nblk + 1) * (size_t)B.rows + sizeof(int32_t) * (size_t)A.rows;
                    size_t free_b = 0, total_b = 0;
                    MI_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
                    // (asked for by the caller's ordering only: not when the bitmaps would crowd the result out of the block cache)
                    if (need < free_b / 3 && (options().spgemm_rank || need <= ((size_t)8 << 30))) {
                        rank_path = true;
                        big.nblk = nblk;
                        big.wpr = wpr;
                        big.bm_store.alloc(sizeof(unsigned) * (size_t)wpr * (size_t)nbig);
                        big.blkcnt.alloc(sizeof(uint16_t) * (size_t)nblk * (size_t)nbig);
                        big.slot_of.alloc(sizeof(int32_t) * (size_t)(A.rows + 1));
                    }
                }
                if (rank_path) {
                    BitmapStore bs;
                    bs.bm = big.bm_store.as<unsigned>();
                    bs.wpr = big.wpr;
                    bs.blkcnt = big.blkcnt.as<uint16_t>();
                    bs.slot_of = big.slot_of.as<int32_t>();
                    bs.nblk = big.nblk;
                    MI_LAUNCH_SMEM((k_spgemm_bitmap<BM_STORE>), dim3((unsigned)nblocks), dim3(1024), bitmap_lds_bytes(B.cols),
                                   c.stream, nbig, (const int32_t*)big_list, B.cols, (const int64_t*)A.ptr,
                                   (const int64_t*)big.ext0.as<int64_t>(), (const int32_t*)big.extlen.as<int32_t>(),
                                   (const int32_t*)B.col, gw, upper, row_nnz,
                                   (const int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, counter, bs);
                    big.have_rank = true;
                } else if (big.b_sorted && n_bounds < ((int64_t)1 << 31)) {
                    big.boff_by_row.alloc(sizeof(int64_t) * (size_t)(A.rows + 1));
                    big.bounds.alloc(sizeof(int32_t) * (size_t)(n_bounds + 1));
                    MI_LAUNCH_SMEM((k_spgemm_bitmap<BM_BOUNDS>), dim3((unsigned)nblocks), dim3(1024), bitmap_lds_bytes(B.cols),
                                   c.stream, nbig, (const int32_t*)big_list, B.cols, (const int64_t*)A.ptr,
                                   (const int64_t*)big.ext0.as<int64_t>(), (const int32_t*)big.extlen.as<int32_t>(),
                                   (const int32_t*)B.col, gw, upper, row_nnz,
                                   (const int64_t*)item_off, big.cap, big.bounds.as<int32_t>(),
                                   big.boff_by_row.as<int64_t>(), counter, BitmapStore{});
                    big.have_bounds = true;
                } else {
                    MI_LAUNCH_SMEM((k_spgemm_bitmap<BM_COUNT>), dim3((unsigned)nblocks), dim3(1024), bitmap_lds_bytes(B.cols),
                                   c.stream, nbig, (const int32_t*)big_list, B.cols, (const int64_t*)A.ptr,
                                   (const int64_t*)big.ext0.as<int64_t>(), (const int32_t*)big.extlen.as<int32_t>(),
                                   (const int32_t*)B.col, gw, upper, row_nnz,
                                   (const int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (int64_t*)nullptr, counter,
                                   BitmapStore{});
                }
            } else if (big.have_rank) {
                // items (runs of blocks of a row) -> one workgroup each (k_spgemm_rank)
                constexpr int CAP = rank_cap<T>();
                if (!big.blkptr.p) {  // where every block starts inside every row of B: B only, built once per symbolic phase
                    big.blkptr.alloc(sizeof(int32_t) * (size_t)(big.nblk + 1) * (size_t)(B.rows + 1));
                    if (B.rows > 0)
                        MI_LAUNCH(k_blkptr, dim3((unsigned)ceil_div(B.rows * WAVE, 256)), dim3(256), c.stream, B.rows,
                                  (const int64_t*)B.ptr, (const int32_t*)B.col, big.nblk, big.blkptr.as<int32_t>());
                }
                const dim3 rgrid((unsigned)ceil_div(nbig, 256));
                MI_LAUNCH(k_rank_items, rgrid, dim3(256), c.stream, nbig, (const int32_t*)big_list,
                          (const int32_t*)big.slot_of.as<int32_t>(), (const uint16_t*)big.blkcnt.as<uint16_t>(), big.nblk, CAP, cptr,
                          (const int64_t*)nullptr, items, (RankItem*)nullptr);
                const int64_t n_items = exclusive_scan_i64(items, item_off, nbig);
                if (options().trace_phases)
                    fprintf(stderr, "[mi_sparse spgemm] big rows %lld, rank items %lld\n", (long long)nbig, (long long)n_items);
                if (n_items) {
                    RankItem* ritems = static_cast<RankItem*>(c.scratch_alloc(sizeof(RankItem) * (size_t)n_items));
                    MI_LAUNCH(k_rank_items, rgrid, dim3(256), c.stream, nbig, (const int32_t*)big_list,
                              (const int32_t*)big.slot_of.as<int32_t>(), (const uint16_t*)big.blkcnt.as<uint16_t>(), big.nblk, CAP,
                              cptr, (const int64_t*)item_off, items, ritems);
                    int64_t* groups_n = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                    int64_t* group_off = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                    MI_LAUNCH(k_rank_groups, rgrid, dim3(256), c.stream, nbig, (const int32_t*)big_list, (const int64_t*)A.ptr,
                              (const int64_t*)item_off, (const int64_t*)nullptr, groups_n, (RankGroup*)nullptr);
                    const int64_t n_groups = exclusive_scan_i64(groups_n, group_off, nbig);
                    RankGroup* rgroups = static_cast<RankGroup*>(c.scratch_alloc(sizeof(RankGroup) * (size_t)(n_groups + 1)));
                    MI_LAUNCH(k_rank_groups, rgrid, dim3(256), c.stream, nbig, (const int32_t*)big_list, (const int64_t*)A.ptr,
                              (const int64_t*)item_off, (const int64_t*)group_off, groups_n, rgroups);
                    const int64_t run8 = RANK_XCD_RUN > 0 ? (int64_t)RANK_XCD_RUN * 8 : 1;
                    launch_batched(ceil_div(n_groups, run8) * run8, RANK_THREADS, [&](int64_t off, int64_t nblk_) {
                        MI_LAUNCH((k_spgemm_rank<T, RANK_THREADS, RANK_UNROLL>), dim3((unsigned)nblk_), dim3(RANK_THREADS), c.stream,
                                  off, n_groups, (const RankGroup*)rgroups, (const RankItem*)ritems,
                                  (const unsigned*)big.bm_store.as<unsigned>(), big.wpr, (const int32_t*)A.col, (const T*)A.val,
                                  (const int64_t*)big.ext0.as<int64_t>(), (const int32_t*)big.blkptr.as<int32_t>(), big.nblk + 1,
                                  (const int32_t*)B.col, (const T*)B.val, ccol, cval, big.col_base);
                    });
                    note_kernel("k_spgemm_rank<%s,%d,%d>", type_name<T>(), RANK_THREADS, RANK_UNROLL);
                }
            } else {
                const int64_t CAP = big.cap;
                MI_LAUNCH(k_part_items, dim3((unsigned)ceil_div(nbig, 256)), dim3(256), c.stream, (const int32_t*)big_list,
                          cnt, nbig, CAP, (int64_t)1 << 62, items);
                const int64_t n_items = exclusive_scan_i64(items, item_off, nbig);
                // slice table (see k_part_slices) unless it would be unreasonably large
                int64_t* n_work = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                int64_t* n_slice = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                int64_t* work_off = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                int64_t* slice_base = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                MI_LAUNCH(k_part_slice_sizes, dim3((unsigned)ceil_div(nbig, 256)), dim3(256), c.stream,
                          (const int32_t*)big_list, (const int64_t*)item_off, (const int64_t*)A.ptr, nbig, n_work, n_slice);
                const int64_t total_work = exclusive_scan_i64(n_work, work_off, nbig);
                const int64_t total_slices = exclusive_scan_i64(n_slice, slice_base, nbig);
                const bool pre = options().spgemm_slice_table && total_slices <= options().spgemm_slice_table_max;
                if (options().trace_phases)
                    fprintf(stderr, "[mi_sparse spgemm] big rows %lld, range items %lld, slice entries %lld, slice threads %lld\n",
                            (long long)nbig, (long long)n_items, (long long)total_slices, (long long)total_work);
                const int32_t* bounds = big.bounds.as<int32_t>();
                const int64_t* brow = big.boff_by_row.as<int64_t>();
                int32_t* bnd = nullptr;
                if (pre) {
                    bnd = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(total_slices + 1)));
                    int32_t* work_t = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(total_work + 1)));
                    if (total_work) {
                        const int64_t wb = ceil_div(total_work, 256);
                        MI_LAUNCH(k_part_item_map, dim3((unsigned)(wb < (1 << 20) ? wb : (1 << 20))), dim3(256), c.stream, total_work,
                                  nbig, (const int64_t*)work_off, work_t);
                    }
                    if (options().trace_phases > 1) { MI_HIP_CHECK(hipStreamSynchronize(c.stream)); fprintf(stderr, "[mi_sparse spgemm]   small bins done\n"); }
                    launch_batched(ceil_div(total_work, 4), 256, [&](int64_t off, int64_t nblk) {  // total_work = waves
                        MI_LAUNCH(k_part_slices, dim3((unsigned)nblk), dim3(256), c.stream, off, (const int32_t*)big_list, nbig,
                                  (const int64_t*)work_off, (const int64_t*)item_off, bounds, brow, (const int64_t*)A.ptr,
                                  (const int32_t*)A.col, (const int64_t*)B.ptr, (const int32_t*)B.col, big.panel_upper ? 1 : upper,
                                  big.diag_shift, (const int64_t*)slice_base, bnd, (const int32_t*)work_t, big.cfloor);
                    });
                }
                if (options().trace_phases > 1) { MI_HIP_CHECK(hipStreamSynchronize(c.stream)); fprintf(stderr, "[mi_sparse spgemm]   slices done\n"); }
                // work items of the numeric kernel: groups of PART_GROUP consecutive ranges of a row
                int64_t* groups = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                int64_t* group_off = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nbig + 1)));
                MI_LAUNCH(k_part_groups, dim3((unsigned)ceil_div(nbig, 256)), dim3(256), c.stream, (const int64_t*)items, nbig,
                          groups);
                const int64_t n_groups = exclusive_scan_i64(groups, group_off, nbig);
                PartDesc* desc = static_cast<PartDesc*>(c.scratch_alloc(sizeof(PartDesc) * (size_t)(nbig + 1)));
                int32_t* item_t = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(n_groups + 1)));
                MI_LAUNCH(k_part_desc, dim3((unsigned)ceil_div(nbig, 256)), dim3(256), c.stream, (const int32_t*)big_list, nbig,
                          (const int64_t*)item_off, (const int64_t*)A.ptr, (const int64_t*)slice_base, cptr, brow,
                          (const int64_t*)group_off, desc);
                if (n_groups) {
                    const int64_t gb = ceil_div(n_groups, 256);
                    MI_LAUNCH(k_part_item_map, dim3((unsigned)(gb < (1 << 20) ? gb : (1 << 20))), dim3(256), c.stream, n_groups,
                              nbig, (const int64_t*)group_off, item_t);
                }
                auto launch = [&](auto log2s_tag, auto pre_tag) {
                    constexpr int L = decltype(log2s_tag)::value;
                    constexpr bool P = decltype(pre_tag)::value;
                    const int64_t run8 = PART_XCD_RUN > 0 ? (int64_t)PART_XCD_RUN * 8 : 1;
                    launch_batched(ceil_div(n_groups, run8) * run8, PART_THREADS, [&](int64_t off, int64_t nblk) {
                        MI_LAUNCH((k_spgemm_part<T, L, P>), dim3((unsigned)nblk), dim3(PART_THREADS), c.stream, off, n_groups,
                                  (const int32_t*)item_t, (const PartDesc*)desc, bounds, B.cols, CAP, (const int32_t*)A.col,
                                  (const T*)A.val, (const int64_t*)B.ptr, (const int32_t*)B.col, (const T*)B.val,
                                  big.panel_upper ? 1 : upper, (const int32_t*)bnd, ccol, cval, big.col_base, big.diag_shift, big.cfloor);
                    });
                };
                if (n_groups) {
                    big.out_range_cap = CAP;
                    constexpr int LO = sizeof(T) >= 16 ? 10 : 11, HI = LO + 1;
                    using lo_t = std::integral_constant<int, LO>;
                    using hi_t = std::integral_constant<int, HI>;
                    if (big.log2s == HI) {
                        if (pre) launch(hi_t{}, std::true_type{}); else launch(hi_t{}, std::false_type{});
                    } else {
                        if (pre) launch(lo_t{}, std::true_type{}); else launch(lo_t{}, std::false_type{});
                    }
                }
            }
        }
    }
    // Rows beyond the LDS bins.  Two forms of the global-memory hash:
    //   * one persistent workgroup per row with a PRIVATE table and workgroup-scope (L2-local) atomics
    //     -- classes up to SPGEMM_COOP_MIN entries;
    //   * cooperative: many workgroups per row on one table with agent-scope atomics -- the hub
    //     classes above that, where a single workgroup per row would serialise.
    if (options().spgemm_global_mode == 0) {
        // global-memory hash: every class that did not go to an LDS bin, largest class first.  One launch
        // per size class so that tables (cleared and compacted per row) and workgroups are sized for
        // the class: many small workgroups for the mid-size rows, few large ones for the hub rows.
        {
            const int first = 0;  // bins already served above have n == 0
            for (int k = NBINS - 1; k >= first; --k) {
                if (!b.n[k]) continue;
                int64_t limit = bin_limit(k);
                if (k == NBINS - 1 || limit > max_cnt) limit = max_cnt;
                const int64_t cap = limit < B.cols ? limit : B.cols;
                int64_t slab = 4;
                while (slab < 2 * cap) slab <<= 1;
                const int threads = cap <= 16384 ? 256 : cap <= 131072 ? 512 : 1024;
                int64_t nblocks = (int64_t)256 * (2048 / threads);
                if (nblocks > b.n[k]) nblocks = b.n[k];
                const size_t per = (size_t)slab * (sizeof(int32_t) + (NUMERIC ? sizeof(T) : 0));
                while (nblocks > 1 && per * (size_t)nblocks > (size_t(4) << 30)) nblocks >>= 1;  // <= 4 GiB of slabs
                unsigned long long* work = static_cast<unsigned long long*>(c.scratch_alloc(sizeof(unsigned long long)));
                MI_HIP_CHECK(hipMemsetAsync(work, 0, sizeof(unsigned long long), c.stream));
                DevBuf kbuf, vbuf;
                kbuf.alloc(sizeof(int32_t) * (size_t)slab * (size_t)nblocks);
                if (NUMERIC) vbuf.alloc(sizeof(T) * (size_t)slab * (size_t)nblocks);
                MI_LAUNCH((k_spgemm_global<T, NUMERIC>), dim3((unsigned)nblocks), dim3(threads), c.stream, b.n[k],
                          (const int32_t*)b.list[k], cnt, B.cols, (const int64_t*)A.ptr, (const int32_t*)A.col,
                          (const T*)A.val, (const int64_t*)B.ptr, (const int32_t*)B.col, (const T*)B.val,
                          gw > threads ? threads : gw, (int)upper, kbuf.as<int32_t>(), vbuf.as<T>(), slab, row_nnz, cptr,
                          ccol, cval, work, big.col_base);
                MI_HIP_CHECK(hipStreamSynchronize(c.stream));  // slabs are freed on scope exit
            }
        }
    } else {
        // global-memory hash: every class that did not go to an LDS bin, largest first.  All rows of a
        // class use one table size; they are processed in batches whose tables fit a 4 GiB workspace.
        {
            const int first = 0;  // bins already served above have n == 0
            for (int k = NBINS - 1; k >= first; --k) {
                if (!b.n[k]) continue;
                int64_t limit = bin_limit(k);
                if (k == NBINS - 1 || limit > max_cnt) limit = max_cnt;
                const int64_t cap = limit < B.cols ? limit : B.cols;
                int log2s = 2;
                while (((int64_t)1 << log2s) < 2 * cap) ++log2s;
                const int64_t S = (int64_t)1 << log2s;
                const size_t per = (size_t)S * (sizeof(int32_t) + (NUMERIC ? sizeof(T) : 0));
                int64_t batch = (int64_t)((size_t(4) << 30) / per);
                if (batch < 1) batch = 1;
                if (batch > 65535) batch = 65535;
                if (batch > b.n[k]) batch = b.n[k];
                DevBuf kbuf, vbuf, cur;
                kbuf.alloc(sizeof(int32_t) * (size_t)S * (size_t)batch);
                if (NUMERIC) {
                    vbuf.alloc(sizeof(T) * (size_t)S * (size_t)batch);
                    cur.alloc(sizeof(unsigned long long) * (size_t)batch);
                }
                const int gwg = gw > GINS_THREADS ? GINS_THREADS : gw;
                for (int64_t r0 = 0; r0 < b.n[k]; r0 += batch) {
                    const int64_t nb = (b.n[k] - r0 < batch) ? b.n[k] - r0 : batch;
                    MI_HIP_CHECK(hipMemsetAsync(kbuf.p, 0xFF, sizeof(int32_t) * (size_t)S * (size_t)nb, c.stream));
                    if (NUMERIC) {
                        MI_HIP_CHECK(hipMemsetAsync(vbuf.p, 0, sizeof(T) * (size_t)S * (size_t)nb, c.stream));
                        MI_HIP_CHECK(hipMemsetAsync(cur.p, 0, sizeof(unsigned long long) * (size_t)nb, c.stream));
                    }
                    // work items of this batch: one per slice of GINS_ENTRIES A-nonzeros
                    int64_t* items = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nb + 1)));
                    int64_t* item_off = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nb + 1)));
                    MI_LAUNCH(k_gins_items, dim3((unsigned)ceil_div(nb, 256)), dim3(256), c.stream,
                              (const int32_t*)(b.list[k] + r0), (const int64_t*)A.ptr, nb, items);
                    const int64_t n_items = exclusive_scan_i64(items, item_off, nb);
                    if (n_items > 2000000000) fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "SpGEMM batch too large for one launch");
                    if (n_items > 0)
                        MI_LAUNCH((k_spgemm_ginsert<T, NUMERIC>), dim3((unsigned)n_items), dim3(GINS_THREADS), c.stream,
                                  (const int32_t*)(b.list[k] + r0), (const int64_t*)A.ptr, (const int32_t*)A.col,
                                  (const T*)A.val, (const int64_t*)B.ptr, (const int32_t*)B.col, (const T*)B.val, gwg,
                                  (int)upper, kbuf.as<int32_t>(), vbuf.as<T>(), log2s, (unsigned long long*)row_nnz,
                                  (const int64_t*)item_off, nb);
                    if (NUMERIC)
                        MI_LAUNCH((k_spgemm_gcompact<T>), dim3((unsigned)ceil_div(S, GCOMP_SLOTS), (unsigned)nb), dim3(256),
                                  c.stream, (const int32_t*)(b.list[k] + r0), (const int32_t*)kbuf.as<int32_t>(),
                                  (const T*)vbuf.as<T>(), log2s, cur.as<unsigned long long>(), cptr, ccol, cval, big.col_base);
                }
                MI_HIP_CHECK(hipStreamSynchronize(c.stream));  // workspaces are freed on scope exit
            }
        }
    }
#undef MI_SPGEMM_ARGS
#undef MI_SPGEMM_LDS_ARGS
    MI_HIP_CHECK(hipGetLastError());
}

// What the symbolic phase computes and the numeric phase consumes.  Kept on the result handle by the staged
// (sp2m-style) API so that the numeric phase can be repeated for new values on an unchanged pattern.
struct SpgemmSymbolic {
    BigRows big;
    DevBuf row_nnz;        // int64[rows + 1]: exact length of every row of C
    int upper_mode = 0;
    int64_t max_nnz = 0;   // longest row of C
    bool done = false;
    // entry-order generations (Csr::order_gen) of the operands big.ext0 / big.extlen were computed for: the tables are indexed
    // by the POSITION of A's entries, so a re-ordered A (mi_sparse_order between the stages) needs them again
    uint64_t a_gen = 0, b_gen = 0;
    int64_t a_nnz = 0, b_nnz = 0;
};

// Phase 0: upper bound of every row of C and, per nonzero of A, the extent of B's row that counts (k_row_ub).
struct SpgemmBounds {
    int64_t* ub = nullptr;  // scratch, A.rows + 1
    int64_t max_ub = 0, sum_ub = 0;
    RowStats stats;         // of ub: sum, maximum, rows per size class
    Bins lists;             // row lists of the symbolic phase (scratch memory: valid inside the API call that built them)
    bool lists_valid = false;
};

static void trace_mark(const char* what, std::chrono::steady_clock::time_point& t_last)
{
    if (!options().trace_phases) return;
    MI_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[mi_sparse spgemm] %-18s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
}

// k_row_ub + k_row_ub_long on the context's stream.  B's row extents are gathered from an int32 copy of its row pointer made
// here (scratch of the call) when nnz(B) < 2^31: half the table, twice the pointers per 128-byte line.
static void launch_row_ub(const Csr& A, const Csr& B, int upper_mode, int64_t* ub, int64_t* ext0, int32_t* extlen,
                          int64_t row_shift = 0)
{
    Context& c = ctx();
    if (A.rows <= 0) return;
    const size_t max_long = (size_t)(A.nnz / ROWUB_LONG + 1);
    int32_t* long_list = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (max_long + 1)));
    unsigned* long_count = reinterpret_cast<unsigned*>(long_list + max_long);
    MI_HIP_CHECK(hipMemsetAsync(long_count, 0, sizeof(unsigned), c.stream));
    const dim3 grid((unsigned)ceil_div(A.rows << ROWUB_LPR_LOG2, 256));
    const unsigned long_grid = (unsigned)std::min<size_t>(max_long, (size_t)8 * (size_t)std::max(c.cus, 1));
    auto run = [&](auto* bptr) {
        using BP = std::remove_cv_t<std::remove_pointer_t<decltype(bptr)>>;
        MI_LAUNCH(k_row_ub<BP>, grid, dim3(256), c.stream, A.rows, (const int64_t*)A.ptr, (const int32_t*)A.col, (const BP*)bptr,
                  (const int32_t*)B.col, upper_mode, ub, ext0, extlen, long_list, long_count, (int32_t)row_shift);
        MI_LAUNCH(k_row_ub_long<BP>, dim3(long_grid), dim3(256), c.stream, (const int32_t*)long_list, (const unsigned*)long_count,
                  (const int64_t*)A.ptr, (const int32_t*)A.col, (const BP*)bptr, (const int32_t*)B.col, upper_mode, ub, ext0, extlen,
                  (int32_t)row_shift);
    };
    if (options().spgemm_narrow_ptr && B.nnz < ((int64_t)1 << 31) - 1 && A.nnz >= ((int64_t)1 << 18) && B.rows >= ((int64_t)1 << 16)) {
        int32_t* bptr32 = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(B.rows + 1)));
        MI_LAUNCH(k_narrow_ptr, dim3((unsigned)std::min<int64_t>(ceil_div(B.rows + 1, 256), 4096)), dim3(256), c.stream,
                  (const int64_t*)B.ptr, B.rows + 1, bptr32);
        run((const int32_t*)bptr32);
    } else {
        run((const int64_t*)B.ptr);
    }
}

template <typename T>
static SpgemmBounds spgemm_bounds(const Csr& A, const Csr& B, bool upper, Csr& C, SpgemmSymbolic& st, int64_t panel_col0 = -1)
{
    Context& c = ctx();
    auto t_last = std::chrono::steady_clock::now();
    if (A.cols != B.rows)
        fail(MI_SPARSE_STATUS_INVALID_VALUE, "dimension mismatch: (%lld x %lld) * (%lld x %lld)", (long long)A.rows,
             (long long)A.cols, (long long)B.rows, (long long)B.cols);
    // upper triangle of a product with sorted B rows: the part of every B row left of the diagonal is skipped
    // by a search instead of being read and dropped (half of the products of a gram matrix)
    BigRows& big = st.big;
    big = BigRows();
    big.b_sorted = rows_sorted(B);
    st.upper_mode = upper ? (big.b_sorted ? 2 : 1) : 0;
    // table of the numeric big-row kernel: 2048 slots (1024 for complex double) give the best occupancy on wide
    // power-law products; a narrow B (dense-ish result rows, few ranges per row) is better off with twice that
    big.log2s = (sizeof(T) >= 16 ? 10 : 11) + (B.cols <= 65536 ? 1 : 0) + (int)options().spgemm_part_log2s_bias;
    if (big.log2s < (sizeof(T) >= 16 ? 10 : 11)) big.log2s = sizeof(T) >= 16 ? 10 : 11;
    if (big.log2s > (sizeof(T) >= 16 ? 11 : 12)) big.log2s = sizeof(T) >= 16 ? 11 : 12;
#ifndef MI_PART_FILL_8THS
#define MI_PART_FILL_8THS 5
#endif
    big.cap = ((int64_t)1 << big.log2s) * MI_PART_FILL_8THS / 8;  // distinct columns per range: the fill of the range kernel's table
    // k_spgemm_bitmap forms ceil(rank / cap) as (rank + cap - 1) * ceil(2^40 / cap) >> 40: exact while (rank + cap) * cap < 2^40, rank <= 2^21 columns
    static_assert((((uint64_t)1 << 21) + ((uint64_t)1 << 12)) * ((uint64_t)1 << 12) < ((uint64_t)1 << 40), "range-start division by multiplication");
    C.rows = A.rows;
    C.cols = B.cols;
    C.ptr_own.alloc(sizeof(int64_t) * (size_t)(C.rows + 1));
    C.ptr = C.ptr_own.as<int64_t>();
    SpgemmBounds bd;
    bd.ub = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(A.rows + 1)));
    big.ext0.alloc(sizeof(int64_t) * (size_t)(A.nnz + 1));
    big.extlen.alloc(sizeof(int32_t) * (size_t)(A.nnz + 1));
    if (options().spgemm_group && B.rows > 0 && B.nnz >= 6 * B.rows) {  // short, even rows of B: 16 lanes per row, no flat list
        if (cache_get(B.gram_max_row) < 0) cache_set(B.gram_max_row, device_max_row_len(B));
        big.grp = cache_get(B.gram_max_row) <= 32;
    }
    if (panel_col0 >= 0 && upper) {
        // B is the column panel starting at column panel_col0 of a wider matrix (spgemm_panels; its rows are sorted): the extents
        // are cut at the shifted diagonal HERE, and the kernels behind them run without the per-product test (whose column
        // indices are the panel's own)
        if (st.upper_mode != 2) fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "column panel with unsorted rows");
        launch_row_ub(A, B, 2, bd.ub, big.ext0.as<int64_t>(), big.extlen.as<int32_t>(), panel_col0);
        st.upper_mode = 0;
        big.panel_upper = true;
        big.diag_shift = (int32_t)panel_col0;
    } else {
        launch_row_ub(A, B, st.upper_mode, bd.ub, big.ext0.as<int64_t>(), big.extlen.as<int32_t>());
    }
    st.a_gen = A.order_gen;
    st.b_gen = B.order_gen;
    st.a_nnz = A.nnz;
    st.b_nnz = B.nnz;
    bd.stats = device_row_stats(bd.ub, A.rows);
    bd.sum_ub = bd.stats.sum;
    bd.max_ub = bd.stats.max;
    trace_mark("row upper bounds", t_last);
    return bd;
}

// One product on the hub path (spgemm_hub.inc): lives for the call.
struct HubState {
    int nbd = 0;          // dense-capable column blocks (relabelled columns [0, bcol0[nbd]))
    int wmax = 0;         // widest of them (accumulators of k_hub_num)
    int nw = 4;           // waves per workgroup of k_hub_sym / k_hub_num: one per 1024 slots of the widest block (8, 4, 2 or 1)
    int64_t nhub = 0;     // rows with at least one dense block, in order of decreasing block count
    int64_t rows1 = 0;    // B.rows + 1: stride of a block's row pointer in ptrb
    int64_t n_items = 0;  // (row, block) pairs
    DevBuf newid, inv;    // int32[cols]: relabelling of B's columns and its inverse
    DevBuf bcol0;         // int32[nbd + 1]: first relabelled column of every block
    DevBuf lut;           // uint16 per granule of HUB_GRAN relabelled columns: its block
    DevBuf ptrb;          // int32[nbd * rows1]: CSR row pointers of the blocks into lcol / lval
    DevBuf lcol, lval;    // uint16 local column, value: the dense-capable entries of B, block-major
    DevBuf hrow;          // HubRow[nhub]
    DevBuf off;           // int64[nbd + 1]: first item of every block
    DevBuf items;         // HubItem[n_items], block-major
    DevBuf dcnt, doff;    // int32[nbd * nhub]: entries of C per (block, hub row) and where they start inside the row's dense part
    DevBuf dtot;          // int64[A.rows + 1]: length of the dense part of every row of C (0: not a hub row)
    DevBuf cfloor;        // int32[A.rows]: first relabelled column of the range path (0: the whole row)
    DevBuf err;           // unsigned: items whose numeric count differed from the symbolic one
    Csr Bs;               // B relabelled, rows sorted
};

// Phase 1: row pointer of C (C.ptr, C.nnz) -- binning, symbolic hash / bitmap kernels, scan.
template <typename T>
static void spgemm_symbolic(const Csr& A, const Csr& B, Csr& C, SpgemmSymbolic& st, SpgemmBounds& bd, HubState* hub = nullptr)
{
    Context& c = ctx();
    auto t_last = std::chrono::steady_clock::now();
    st.row_nnz.alloc(sizeof(int64_t) * (size_t)(A.rows + 1));
    int64_t* row_nnz = st.row_nnz.as<int64_t>();
    MI_HIP_CHECK(hipMemsetAsync(row_nnz, 0, sizeof(int64_t) * (size_t)(A.rows + 1), c.stream));
    run_phase<T, false>(A, B, st.upper_mode, bd.ub, bd.stats, row_nnz, nullptr, nullptr, nullptr, st.big, &bd.lists);
    bd.lists_valid = bd.max_ub <= 2048;  // every row in an LDS class of every value type (classes 0-6)
    trace_mark("symbolic", t_last);
    if (hub) {
        // dense blocks of the hub rows: distinct columns per (row, block), their places inside the row, the rows' totals
        HubState& h = *hub;
        {
            const unsigned grid_s = (unsigned)std::min<int64_t>(h.n_items, (int64_t)c.cus * (32 / h.nw));
            auto go = [&](auto nw_tag) {
                constexpr int NW = decltype(nw_tag)::value;
                MI_LAUNCH((k_hub_sym<NW>), dim3(grid_s), dim3(NW * 64), c.stream, h.n_items, (const HubItem*)h.items.as<HubItem>(), h.nhub,
                          (const int32_t*)A.col, (const int32_t*)h.ptrb.as<int32_t>(), h.rows1, (const uint16_t*)h.lcol.as<uint16_t>(),
                          (const int32_t*)h.bcol0.as<int32_t>(), h.dcnt.as<int32_t>());
            };
            if (h.nw == 8) go(std::integral_constant<int, 8>{});
            else if (h.nw == 4) go(std::integral_constant<int, 4>{});
            else if (h.nw == 2) go(std::integral_constant<int, 2>{});
            else go(std::integral_constant<int, 1>{});
        }
        MI_LAUNCH(k_hub_offsets, dim3((unsigned)ceil_div(h.nhub, 256)), dim3(256), c.stream, h.nhub, (const HubRow*)h.hrow.as<HubRow>(),
                  (const int32_t*)h.dcnt.as<int32_t>(), h.doff.as<int32_t>(), h.dtot.as<int64_t>());
        trace_mark("hub symbolic", t_last);
        // a row of C = [range-path part: row_nnz][dense part: dtot]
        int64_t* total = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(A.rows + 1)));
        MI_LAUNCH(k_hub_add_len, dim3((unsigned)ceil_div(A.rows, 256)), dim3(256), c.stream, A.rows, (const int64_t*)row_nnz,
                  (const int64_t*)h.dtot.as<int64_t>(), total);
        C.nnz = exclusive_scan_i64(total, C.ptr, C.rows);
        MI_LAUNCH(k_hub_out0, dim3((unsigned)ceil_div(h.nhub, 256)), dim3(256), c.stream, h.nhub, h.hrow.as<HubRow>(),
                  (const int64_t*)C.ptr, (const int64_t*)row_nnz);
    } else
    C.nnz = exclusive_scan_i64(row_nnz, C.ptr, C.rows);
    C.col = nullptr;
    C.val = nullptr;
    C.valid = false;
    st.done = true;
    trace_mark("scan", t_last);
}

// The same product with the row of C built in LDS (round 5): one workgroup per output row walks the row's products once per
// column tile of 128 KiB, adds them into the tile (LDS atomics: no global atomic, no zero fill of C beforehand) and writes the
// tile once.  k_fill_dense + k_spmmd wrote the 2 GiB result of two 2^14-square operands twice and went through global atomics:
// 1.39 ms.  Rows of C wider than a tile re-walk the row's products per tile (the dense form is for small M N: SURVEY a4).
constexpr int SPMMD_TILE_BYTES = 128 * 1024;
template <typename T>
__global__ void __launch_bounds__(1024)
    k_spmmd_lds(int64_t rows, int64_t ncols, const int64_t* __restrict__ aptr, const int32_t* __restrict__ acol,
                const T* __restrict__ aval, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol,
                const T* __restrict__ bval, T* __restrict__ C, int64_t c_rs, int64_t c_cs)
{
    constexpr int TW = SPMMD_TILE_BYTES / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) T acc[TW];
    const int tid = threadIdx.x, wave = tid / WAVE, lane = tid % WAVE, nwaves = blockDim.x / WAVE;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int64_t a0 = aptr[row], a1 = aptr[row + 1];
        T* crow = C + row * c_rs;
        for (int64_t j0 = 0; j0 < ncols; j0 += TW) {
            const int64_t j1 = j0 + TW < ncols ? j0 + TW : ncols;
            for (int k = tid; k < (int)(j1 - j0); k += blockDim.x) acc[k] = vt<T>::zero();
            __syncthreads();
            for (int64_t p = a0 + wave; p < a1; p += nwaves) {
                const int32_t kk = acol[p];
                const T a = aval[p];
                for (int64_t q = bptr[kk] + lane; q < bptr[kk + 1]; q += WAVE) {
                    const int64_t j = bcol[q];
                    if (j >= j0 && j < j1) lds_accum(&acc[j - j0], vt<T>::mul(a, bval[q]));
                }
            }
            __syncthreads();
            for (int k = tid; k < (int)(j1 - j0); k += blockDim.x) crow[(j0 + k) * c_cs] = acc[k];
            __syncthreads();
        }
    }
}

// option "deterministic": the hash kernels add the products of an entry in whatever order their waves arrive: keep their
// PATTERN, put the columns of every row in order (a unique arrangement) and form the values again in a fixed order
template <typename T>
static void spgemm_values_deterministic(const Csr& A, const Csr& B, Csr& C, int upper_mode)
{
    Context& c = ctx();
    sort_csr(type_char<T>::value, C);
    MI_HIP_CHECK(hipMemsetAsync(C.val, 0, sizeof(T) * (size_t)C.nnz, c.stream));
    constexpr int RPB = 256 / WAVE;  // rows per workgroup
    launch_batched(ceil_div(A.rows, (int64_t)RPB), 256, [&](int64_t off, int64_t nb) {
        const int64_t r0 = off * RPB;
        MI_LAUNCH((k_spgemm_values_det<T>), dim3((unsigned)nb), dim3(256), c.stream, r0, A.rows, (const int64_t*)A.ptr,
                  (const int32_t*)A.col, (const T*)A.val, (const int64_t*)B.ptr, (const int32_t*)B.col, (const T*)B.val,
                  upper_mode != 0 ? 1 : 0, (const int64_t*)C.ptr, (const int32_t*)C.col, static_cast<T*>(C.val));
    });
}

// rows (= waves) per workgroup of the one-pass kernel: every workgroup draws a ticket with ONE atomic on ONE address, which the
// memory side performs at ~80 M / s -- with four rows per ticket the counter, not the product, set the pace (round 4:
// 1 / 2 / 4 rows per workgroup = 12.7 / 6.6 / 3.4 ms on the uniform configs[2], exactly 12 ns per ticket)
#ifndef MI_ONEPASS_WAVES
#define MI_ONEPASS_WAVES 8
#endif
constexpr int onepass_waves(int log2s, size_t vbytes)
{
    int w = MI_ONEPASS_WAVES;
    while (w > 1 && (size_t)w * ((size_t)1 << log2s) * (sizeof(int32_t) + vbytes) > (size_t)96 * 1024) w >>= 1;
    return w;
}

// One pass (k_spgemm_onepass): taken when every row has at most 512 products and the upper bound of nnz(C) is affordable.
// Returns false (nothing done) otherwise.
template <typename T>
static bool spgemm_onepass(const Csr& A, const Csr& B, Csr& C, SpgemmSymbolic& st, const SpgemmBounds& bd)
{
    Context& c = ctx();
    if (!options().spgemm_onepass || options().spgemm_force_global) return false;
    // every workgroup draws its ticket from ONE counter (~12 ns each at the memory side) and rows retire in order: measured
    // faster than two phases up to ~10^5 rows (fewer launches and host round trips: 2^14 x 2^14, 16 / row: 0.30 -> 0.17 ms)
    // and slower beyond (2^20 rows: 2.9 -> 4.6 ms; 3.4 ms without any look-back, 4.1-4.5 ms with block indices / persistent
    // workgroups instead of tickets: profiles/r04_spgemm_onepass_variants.log); option value 2 forces it
    if (options().spgemm_onepass == 1 && A.rows > 65536) return false;
    if (A.rows < 1 || bd.max_ub > 512 || bd.sum_ub < 1) return false;
    const size_t per = sizeof(int32_t) + sizeof(T);
    size_t free_b = 0, total_b = 0;
    MI_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    if ((size_t)bd.sum_ub * per > total_b / 4) return false;  // the arrays are sized for the bound, not for nnz(C)
    auto t_last = std::chrono::steady_clock::now();
    int log2s = 10;
    if (bd.max_ub <= 32) log2s = 6;
    else if (bd.max_ub <= 128) log2s = 8;
    else if (bd.max_ub <= 256) log2s = 9;
    const int waves = onepass_waves(log2s, sizeof(T));
    const int64_t nblocks = ceil_div(A.rows, (int64_t)waves);
    unsigned long long* flags = static_cast<unsigned long long*>(c.scratch_alloc(sizeof(unsigned long long) * (size_t)(nblocks + 1)));
    MI_HIP_CHECK(hipMemsetAsync(flags, 0, sizeof(unsigned long long) * (size_t)(nblocks + 1), c.stream));
    unsigned long long* ticket = flags + nblocks;
    try {  // sized for the bound: a device that is nearly full may still hold the exact result of the two-phase path
        C.col_own.alloc(sizeof(int32_t) * (size_t)bd.sum_ub);
        C.val_own.alloc(sizeof(T) * (size_t)bd.sum_ub);
    } catch (const status_error& e) {
        if (e.status != MI_SPARSE_STATUS_ALLOC_FAILED) throw;
        C.col_own.release();
        C.val_own.release();
        clear_error();
        return false;
    }
    C.col = C.col_own.as<int32_t>();
    C.val = C.val_own.p;
    auto launch = [&](auto log2s_tag) {
        constexpr int L = decltype(log2s_tag)::value;
        constexpr int WAVES = onepass_waves(L, sizeof(T));
        launch_batched(nblocks, WAVES * 64, [&](int64_t, int64_t nb) {  // tickets, not block indices, pick the rows
            MI_LAUNCH((k_spgemm_onepass<T, L, WAVES>), dim3((unsigned)nb), dim3(WAVES * 64), c.stream, A.rows,
                      (const int64_t*)A.ptr, (const int64_t*)bd.ub, (const int64_t*)st.big.ext0.as<int64_t>(),
                      (const int32_t*)st.big.extlen.as<int32_t>(), (const T*)A.val, (const int32_t*)B.col, (const T*)B.val,
                      st.upper_mode, ticket, flags, C.ptr, C.col, static_cast<T*>(C.val));
        });
        note_kernel("k_spgemm_onepass<%s,%d,%d>", type_name<T>(), L, WAVES);
    };
    if (log2s == 6) launch(std::integral_constant<int, 6>{});
    else if (log2s == 8) launch(std::integral_constant<int, 8>{});
    else if (log2s == 9) launch(std::integral_constant<int, 9>{});
    else launch(std::integral_constant<int, 10>{});
    MI_HIP_CHECK(hipGetLastError());
    MI_HIP_CHECK(hipMemcpyAsync(&C.nnz, C.ptr + A.rows, sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    MI_HIP_CHECK(hipStreamSynchronize(c.stream));
    trace_mark("one pass", t_last);
    if (C.nnz * 2 < bd.sum_ub || (size_t)(bd.sum_ub - C.nnz) * per > ((size_t)1 << 30)) {  // the bound was loose (or a GiB over): give the surplus back
        DevBuf col, val;
        col.alloc(sizeof(int32_t) * (size_t)C.nnz);
        val.alloc(sizeof(T) * (size_t)C.nnz);
        if (C.nnz > 0) {
            MI_HIP_CHECK(hipMemcpyAsync(col.p, C.col, sizeof(int32_t) * (size_t)C.nnz, hipMemcpyDeviceToDevice, c.stream));
            MI_HIP_CHECK(hipMemcpyAsync(val.p, C.val, sizeof(T) * (size_t)C.nnz, hipMemcpyDeviceToDevice, c.stream));
        }
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        C.col_own = std::move(col);
        C.val_own = std::move(val);
        C.col = C.col_own.as<int32_t>();
        C.val = C.val_own.p;
    }
    C.valid = true;
    C.order_gen = next_order_gen();
    C.sorted = false;
    if (options().deterministic && C.nnz > 0) spgemm_values_deterministic<T>(A, B, C, st.upper_mode);
    return true;
}

// Phase 2: column indices and values of C (storage allocated on the first run; repeatable).
template <typename T>
static void spgemm_numeric(const Csr& A, const Csr& B, Csr& C, SpgemmSymbolic& st, SpgemmBounds* bd = nullptr, HubState* hub = nullptr)
{
    Context& c = ctx();
    if (!st.done) fail(MI_SPARSE_STATUS_INVALID_VALUE, "numeric SpGEMM phase requested before the symbolic one");
    if (A.nnz != st.a_nnz || B.nnz != st.b_nnz)
        fail(MI_SPARSE_STATUS_INVALID_VALUE, "operand structure changed since the symbolic phase (nnz %lld x %lld, was %lld x %lld)",
             (long long)A.nnz, (long long)B.nnz, (long long)st.a_nnz, (long long)st.b_nnz);
    if (A.order_gen != st.a_gen || B.order_gen != st.b_gen) {
        // the entries of an operand moved since the symbolic phase (mi_sparse_order between NNZ_COUNT and FINALIZE): the pattern
        // of C, its row lengths and the range starts are unaffected, but the per-entry extents of B's rows follow A's storage
        // order (and, upper triangle of a sorted B, B's) -- one streamed pass rebuilds them.  A B that was unsorted at the
        // symbolic phase keeps mode 1 (every product tested), which is valid for any order.
        int64_t* ub = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(A.rows + 1)));
        launch_row_ub(A, B, st.upper_mode, ub, st.big.ext0.as<int64_t>(), st.big.extlen.as<int32_t>());
        st.a_gen = A.order_gen;
        st.b_gen = B.order_gen;
    }
    if (!C.col_own.p || C.col_own.bytes < sizeof(int32_t) * (size_t)C.nnz) C.col_own.alloc(sizeof(int32_t) * (size_t)C.nnz);
    if (!C.val_own.p || C.val_own.bytes < sizeof(T) * (size_t)C.nnz) C.val_own.alloc(sizeof(T) * (size_t)C.nnz);
    C.col = C.col_own.as<int32_t>();
    C.val = C.val_own.p;
    auto t_last = std::chrono::steady_clock::now();
    if (C.nnz > 0) {
        if (bd && bd->lists_valid) {  // same API call as the symbolic phase, small rows only: its row lists serve again
            run_phase<T, true>(A, B, st.upper_mode, st.row_nnz.as<int64_t>(), bd->stats, nullptr, C.ptr, C.col,
                               static_cast<T*>(C.val), st.big, &bd->lists);
        } else {
            const RowStats rs = device_row_stats(st.row_nnz.as<int64_t>(), A.rows);
            run_phase<T, true>(A, B, st.upper_mode, st.row_nnz.as<int64_t>(), rs, nullptr, C.ptr, C.col,
                               static_cast<T*>(C.val), st.big);
        }
    }
    trace_mark("numeric", t_last);
    if (hub && C.nnz > 0) {
        HubState& h = *hub;
        MI_HIP_CHECK(hipMemsetAsync(h.err.p, 0, sizeof(unsigned), c.stream));
        const size_t smem = sizeof(T) * (size_t)h.wmax;
        // persistent workgroups: as many as the accumulators let a CU hold
        auto go = [&](auto nw_tag) {
            constexpr int NW = decltype(nw_tag)::value;
            const int64_t per_cu = std::max<int64_t>(1, std::min<int64_t>(32 / NW, (int64_t)(156 * 1024) / (int64_t)(smem + NW * sizeof(HubStage<T>) + 256)));
            if (smem > 32 * 1024)  // more than 64 KiB of LDS per workgroup (static + dynamic) needs the attribute
                MI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hub_num<T, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            MI_LAUNCH_SMEM((k_hub_num<T, NW>), dim3((unsigned)std::min<int64_t>(h.n_items, (int64_t)c.cus * per_cu)), dim3(NW * 64), smem,
                           c.stream, h.n_items, (const HubItem*)h.items.as<HubItem>(), (const HubRow*)h.hrow.as<HubRow>(), h.nhub,
                           (const int32_t*)A.col, (const T*)A.val, (const int32_t*)h.ptrb.as<int32_t>(), h.rows1,
                           (const uint16_t*)h.lcol.as<uint16_t>(), (const T*)h.lval.as<T>(), (const int32_t*)h.bcol0.as<int32_t>(),
                           (const int32_t*)h.inv.as<int32_t>(), (const int32_t*)h.dcnt.as<int32_t>(), (const int32_t*)h.doff.as<int32_t>(),
                           C.col, static_cast<T*>(C.val), h.err.as<unsigned>(), h.wmax);
        };
        if (h.nw == 8) go(std::integral_constant<int, 8>{});
        else if (h.nw == 4) go(std::integral_constant<int, 4>{});
        else if (h.nw == 2) go(std::integral_constant<int, 2>{});
        else go(std::integral_constant<int, 1>{});
        note_kernel("k_hub_num<%s> + range path", type_name<T>());
        unsigned bad = 0;
        MI_HIP_CHECK(hipMemcpyAsync(&bad, h.err.p, sizeof(unsigned), hipMemcpyDeviceToHost, c.stream));
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        if (bad) fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "hub path: %u (row, block) items counted differently by the two phases", bad);
        trace_mark("hub numeric", t_last);
    }
    if (options().trace_phases) {
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        fprintf(stderr, "[mi_sparse spgemm] numeric done\n");
    }
    C.valid = true;
    C.order_gen = next_order_gen();
    C.sorted = false;
    // what mi_sparse_order may rely on (handle.hip, k_sort_ranges): rows longer than the LDS classes were written range by
    // range -- consecutive runs of `cap` entries whose column sets are disjoint and ascending from run to run
    C.range_cap = hub ? 0 : st.big.out_range_cap;  // (hub path: a long row is [ranges][dense blocks], and the ranges ascend in the RELABELLED column order)
    C.range_min_len = bin_limit((sizeof(T) >= 16 ? 7 : 8) - 1);
    C.sorted_min_len = st.big.have_rank ? C.range_min_len : 0;  // k_spgemm_rank wrote every longer row in column order
    if (options().deterministic && C.nnz > 0) {
        spgemm_values_deterministic<T>(A, B, C, st.upper_mode);
        C.range_cap = 0;  // (that pass orders the rows itself)
        C.sorted_min_len = 0;
    }
}


// ---- hub path (spgemm_hub.inc): relabel B's columns by popularity, dense LDS accumulators for the leading blocks of the hub
// rows, the range path for everything else.  `bd` / `st` come from spgemm_bounds(A, B) (full product: the extents are whole
// rows of B, valid for the relabelled copy too -- it shares B's row pointer).  false: declined, nothing changed.
template <typename T>
static bool spgemm_hub(const Csr& A, const Csr& B, Csr& C, SpgemmSymbolic& st, SpgemmBounds& bd)
{
    Context& c = ctx();
    const Options& o = options();
    constexpr int first_big = sizeof(T) >= 16 ? 7 : 8;
    const int64_t hub_min = bin_limit(first_big - 1);  // rows of more products are beyond the LDS hash classes
    if (!o.spgemm_hub || o.spgemm_force_global || !o.spgemm_lds_parts || o.spgemm_rank) return false;
    if (bd.max_ub <= hub_min || B.nnz <= 0 || B.nnz >= ((int64_t)1 << 31) - 1 || A.nnz >= ((int64_t)1 << 31) - 1) return false;
    if (bitmap_lds_bytes(B.cols) > (size_t)140 * 1024 || B.cols < 2 * HUB_GRAN) return false;  // (the range path's own limit)
    if ((double)A.rows * (double)ceil_div(B.cols, st.big.cap) >= 2147483647.0) return false;   // its range-start table
    // worth it for products with real hubs only: the relabelled copies of B cost a few milliseconds
    if (o.spgemm_hub == 1 && (bd.sum_ub < o.spgemm_hub_min_products || bd.max_ub < 8 * hub_min)) return false;
    auto t_last = std::chrono::steady_clock::now();
    auto h = std::make_unique<HubState>();
    const int64_t cols = B.cols, rowsB = B.rows;
    const int64_t ngran = ceil_div(cols, (int64_t)HUB_GRAN);
    try {
        // 1. popularity of B's columns -> relabelling (most popular first), entries per granule of new columns
        unsigned* hist = static_cast<unsigned*>(c.scratch_alloc(sizeof(unsigned) * (size_t)(cols + 1)));
        unsigned* ccount = static_cast<unsigned*>(c.scratch_alloc(sizeof(unsigned) * 512));
        unsigned* ccursor = ccount + 256;
        unsigned* gran = static_cast<unsigned*>(c.scratch_alloc(sizeof(unsigned) * (size_t)(ngran + 1)));
        MI_HIP_CHECK(hipMemsetAsync(hist, 0, sizeof(unsigned) * (size_t)(cols + 1), c.stream));
        MI_HIP_CHECK(hipMemsetAsync(ccount, 0, sizeof(unsigned) * 512, c.stream));
        MI_HIP_CHECK(hipMemsetAsync(gran, 0, sizeof(unsigned) * (size_t)(ngran + 1), c.stream));
        h->newid.alloc(sizeof(int32_t) * (size_t)cols);
        h->inv.alloc(sizeof(int32_t) * (size_t)cols);
        const unsigned egrid = (unsigned)std::min<int64_t>(ceil_div(B.nnz, 256), (int64_t)1 << 16);
        const unsigned cgrid = (unsigned)std::min<int64_t>(ceil_div(cols, 256), (int64_t)1 << 14);
        MI_LAUNCH(k_hub_colhist, dim3(egrid), dim3(256), c.stream, B.nnz, (const int32_t*)B.col, hist);
        MI_LAUNCH(k_hub_class_count, dim3(cgrid), dim3(256), c.stream, cols, (const unsigned*)hist, ccount);
        MI_LAUNCH(k_hub_class_scan, dim3(1), dim3(256), c.stream, (const unsigned*)ccount, ccursor);
        MI_LAUNCH(k_hub_assign, dim3((unsigned)ceil_div(cols, 256)), dim3(256), c.stream, cols, (const unsigned*)hist, ccursor,
                  h->newid.as<int32_t>(), h->inv.as<int32_t>(), gran);
        std::vector<unsigned> hgran((size_t)ngran);
        MI_HIP_CHECK(hipMemcpyAsync(hgran.data(), gran, sizeof(unsigned) * (size_t)ngran, hipMemcpyDeviceToHost, c.stream));
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        // 2. column blocks: at most wmax columns (the accumulators of one workgroup) and at most ent_max entries of B (what one
        //    block's slices may take of an L2) each; block b is dense for a row of ub products when ub * share_b >= thr * width_b
        int wmax = (int)((o.spgemm_hub_acc_kb * 1024 / (int64_t)sizeof(T)) / HUB_GRAN * HUB_GRAN);
        if (wmax > HUB_WMAX) wmax = HUB_WMAX;
        if (wmax < HUB_GRAN) wmax = HUB_GRAN;
        const int64_t ent_max = std::max<int64_t>(o.spgemm_hub_block_kb * 1024 / (int64_t)(2 + sizeof(T)), 1);
        const double thr = (double)o.spgemm_hub_fill_pct / 100.0;
        std::vector<int32_t> bcol0{0};
        std::vector<double> tmin;
        {
            int64_t ent = 0;
            int32_t start = 0;
            for (int64_t g = 0; g < ngran && (int)tmin.size() < HUB_MAX_BLOCKS; ++g) {
                const int32_t gend = (int32_t)std::min<int64_t>((g + 1) * HUB_GRAN, cols);
                ent += hgran[(size_t)g];
                const int32_t width = gend - start;
                if (width >= wmax || ent >= ent_max || g + 1 == ngran) {
                    const double share = (double)ent / (double)B.nnz;
                    const double t = share > 0.0 ? thr * (double)width / share : 1e300;
                    if ((double)bd.max_ub < t) break;  // not even the heaviest row is dense here (and the blocks only get emptier)
                    bcol0.push_back(gend);
                    tmin.push_back(t);
                    start = gend;
                    ent = 0;
                }
            }
        }
        if (o.spgemm_hub == 3) {
            // experiment: the relabelling alone -- the range path on the relabelled, sorted copy of B (popular columns adjacent:
            // a range of 1280 distinct columns is a narrow stripe of B there, its slices of B's rows are long)
            Csr& Bs = h->Bs;
            Bs.rows = B.rows;
            Bs.cols = B.cols;
            Bs.nnz = B.nnz;
            Bs.ptr = B.ptr;
            Bs.col_own.alloc(sizeof(int32_t) * (size_t)B.nnz);
            Bs.val_own.alloc(sizeof(T) * (size_t)B.nnz);
            Bs.col = Bs.col_own.as<int32_t>();
            Bs.val = Bs.val_own.p;
            MI_LAUNCH(k_hub_relabel, dim3(egrid), dim3(256), c.stream, B.nnz, (const int32_t*)B.col, (const int32_t*)h->newid.as<int32_t>(), Bs.col);
            MI_HIP_CHECK(hipMemcpyAsync(Bs.val, B.val, sizeof(T) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
            Bs.valid = true;
            Bs.sorted = false;
            sort_csr(type_char<T>::value, Bs);
            cache_set(Bs.sorted, true);
            trace_mark("relabelled copy of B", t_last);
            st.big.b_sorted = true;
            st.b_gen = Bs.order_gen;
            st.big.want_rank = false;
            spgemm_symbolic<T>(A, Bs, C, st, bd);
            spgemm_numeric<T>(A, Bs, C, st, &bd);
            if (C.nnz > 0)  // the whole result is in relabelled columns
                MI_LAUNCH(k_hub_unmap, dim3((unsigned)std::min<int64_t>(ceil_div(C.rows, 4), (int64_t)1 << 20)), dim3(256), c.stream, C.rows,
                          (const int64_t*)C.ptr, (const int64_t*)nullptr, (const int32_t*)h->inv.as<int32_t>(), C.col);
            C.range_cap = 0;
            c.sync();
            return true;
        }
        const int nbd = (int)tmin.size();
        if (nbd == 0) return false;
        h->nbd = nbd;
        h->wmax = 0;
        for (int b = 0; b < nbd; ++b) h->wmax = std::max(h->wmax, (int)(bcol0[(size_t)b + 1] - bcol0[(size_t)b]));
        h->nw = h->wmax > 4096 ? 8 : h->wmax > 2048 ? 4 : h->wmax > 1024 ? 2 : 1;
        h->rows1 = rowsB + 1;
        const int32_t cmax = bcol0[(size_t)nbd];
        std::vector<uint16_t> lut((size_t)ceil_div(cmax, (int64_t)HUB_GRAN) + 1, 0);
        for (int b = 0; b < nbd; ++b)
            for (int32_t g = bcol0[(size_t)b] >> HUB_GRAN_LOG2; g < (bcol0[(size_t)b + 1] + HUB_GRAN - 1) >> HUB_GRAN_LOG2; ++g) lut[(size_t)g] = (uint16_t)b;
        // the block table must be affordable next to the result
        if ((double)nbd * (double)h->rows1 * 4.0 > (double)device_total_bytes() / 32.0) return false;
        h->bcol0.alloc(sizeof(int32_t) * (size_t)(nbd + 1));
        h->lut.alloc(lut.size() * sizeof(uint16_t));
        double* tmin_d = static_cast<double*>(c.scratch_alloc(sizeof(double) * (size_t)nbd));
        MI_HIP_CHECK(hipMemcpyAsync(h->bcol0.p, bcol0.data(), sizeof(int32_t) * (size_t)(nbd + 1), hipMemcpyHostToDevice, c.stream));
        MI_HIP_CHECK(hipMemcpyAsync(h->lut.p, lut.data(), lut.size() * sizeof(uint16_t), hipMemcpyHostToDevice, c.stream));
        MI_HIP_CHECK(hipMemcpyAsync(tmin_d, tmin.data(), sizeof(double) * (size_t)nbd, hipMemcpyHostToDevice, c.stream));
        // 3. hub rows: how many dense blocks each, in order of decreasing count; floors
        int32_t* nd_of_row = static_cast<int32_t*>(c.scratch_alloc(sizeof(int32_t) * (size_t)(A.rows + 1)));
        unsigned* ndcount = static_cast<unsigned*>(c.scratch_alloc(sizeof(unsigned) * (size_t)(nbd + 2)));
        MI_HIP_CHECK(hipMemsetAsync(ndcount, 0, sizeof(unsigned) * (size_t)(nbd + 2), c.stream));
        const unsigned rgrid = (unsigned)ceil_div(A.rows, 256);
        MI_LAUNCH(k_hub_nd, dim3(rgrid), dim3(256), c.stream, A.rows, (const int64_t*)bd.ub, hub_min, nbd, (const double*)tmin_d,
                  nd_of_row, ndcount);
        std::vector<unsigned> hnd((size_t)nbd + 2);
        MI_HIP_CHECK(hipMemcpyAsync(hnd.data(), ndcount, sizeof(unsigned) * (size_t)(nbd + 2), hipMemcpyDeviceToHost, c.stream));
        MI_HIP_CHECK(hipStreamSynchronize(c.stream));
        // rows with nd == m start at the number of rows with nd > m; block b is owned by the rows with nd > b
        std::vector<unsigned> start((size_t)nbd + 2, 0);
        std::vector<int64_t> off((size_t)nbd + 1, 0);
        {
            unsigned run = 0;
            for (int m = nbd; m >= 1; --m) {
                start[(size_t)m] = run;
                run += hnd[(size_t)m];
            }
            h->nhub = run;
            int64_t items = 0;
            unsigned above = 0;  // rows with nd > b
            std::vector<unsigned> owners((size_t)nbd, 0);
            for (int b = nbd - 1; b >= 0; --b) {
                above += hnd[(size_t)b + 1];
                owners[(size_t)b] = above;
            }
            for (int b = 0; b < nbd; ++b) {
                off[(size_t)b] = items;
                items += owners[(size_t)b];
            }
            off[(size_t)nbd] = items;
            h->n_items = items;
        }
        if (h->nhub == 0 || h->n_items == 0) return false;
        if ((double)nbd * (double)h->nhub * 8.0 > (double)device_total_bytes() / 32.0) return false;
        MI_HIP_CHECK(hipMemcpyAsync(ndcount, start.data(), sizeof(unsigned) * (size_t)(nbd + 2), hipMemcpyHostToDevice, c.stream));
        h->off.alloc(sizeof(int64_t) * (size_t)(nbd + 1));
        MI_HIP_CHECK(hipMemcpyAsync(h->off.p, off.data(), sizeof(int64_t) * (size_t)(nbd + 1), hipMemcpyHostToDevice, c.stream));
        h->hrow.alloc(sizeof(HubRow) * (size_t)h->nhub);
        h->cfloor.alloc(sizeof(int32_t) * (size_t)(A.rows + 1));
        h->dtot.alloc(sizeof(int64_t) * (size_t)(A.rows + 1));
        h->err.alloc(sizeof(unsigned));
        MI_HIP_CHECK(hipMemsetAsync(h->dtot.p, 0, sizeof(int64_t) * (size_t)(A.rows + 1), c.stream));
        MI_LAUNCH(k_hub_rank, dim3(rgrid), dim3(256), c.stream, A.rows, (const int32_t*)nd_of_row, ndcount,
                  (const int32_t*)h->bcol0.as<int32_t>(), (const int64_t*)A.ptr, h->cfloor.as<int32_t>(), h->hrow.as<HubRow>());
        h->items.alloc(sizeof(HubItem) * (size_t)h->n_items);
        MI_LAUNCH(k_hub_items, dim3((unsigned)std::min<int64_t>(ceil_div(h->n_items, 256), (int64_t)1 << 16)), dim3(256), c.stream, h->n_items,
                  (const int64_t*)h->off.as<int64_t>(), nbd, (const HubRow*)h->hrow.as<HubRow>(), h->items.as<HubItem>());
        trace_mark("hub: relabel, rows", t_last);
        // 4. B relabelled, rows sorted (shares B's row pointer)
        Csr& Bs = h->Bs;
        Bs.rows = B.rows;
        Bs.cols = B.cols;
        Bs.nnz = B.nnz;
        Bs.ptr = B.ptr;
        Bs.col_own.alloc(sizeof(int32_t) * (size_t)B.nnz);
        Bs.val_own.alloc(sizeof(T) * (size_t)B.nnz);
        Bs.col = Bs.col_own.as<int32_t>();
        Bs.val = Bs.val_own.p;
        MI_LAUNCH(k_hub_relabel, dim3(egrid), dim3(256), c.stream, B.nnz, (const int32_t*)B.col, (const int32_t*)h->newid.as<int32_t>(), Bs.col);
        MI_HIP_CHECK(hipMemcpyAsync(Bs.val, B.val, sizeof(T) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
        Bs.valid = true;
        Bs.sorted = false;
        sort_csr(type_char<T>::value, Bs);
        cache_set(Bs.sorted, true);
        trace_mark("hub: sorted copy of B", t_last);
        // 5. its dense-capable columns block-major
        {
            DevBuf rowid_b;
            rowid_b.alloc(sizeof(int32_t) * (size_t)B.nnz);
            MI_LAUNCH(k_hub_rowid, dim3(egrid), dim3(256), c.stream, B.nnz, (const int64_t*)Bs.ptr, Bs.rows, rowid_b.as<int32_t>());
            const int64_t ntab = (int64_t)nbd * h->rows1;
            h->ptrb.alloc(sizeof(int32_t) * (size_t)(ntab + 1));
            MI_HIP_CHECK(hipMemsetAsync(h->ptrb.p, 0, sizeof(int32_t) * (size_t)(ntab + 1), c.stream));
            MI_LAUNCH(k_hub_blk_count, dim3(egrid), dim3(256), c.stream, B.nnz, (const int32_t*)rowid_b.as<int32_t>(), (const int64_t*)Bs.ptr,
                      (const int32_t*)Bs.col, cmax, (const uint16_t*)h->lut.as<uint16_t>(), (const int32_t*)h->bcol0.as<int32_t>(), h->rows1,
                      h->ptrb.as<int32_t>());
            const int64_t ntiles = ceil_div(ntab + 1, (int64_t)HUB_SCAN_TILE);
            int64_t* sums = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(ntiles + 1)));
            int64_t* offs = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(ntiles + 1)));
            MI_LAUNCH(k_hub_scan_sums, dim3((unsigned)ntiles), dim3(256), c.stream, (const int32_t*)h->ptrb.as<int32_t>(), ntab + 1, sums);
            const int64_t nblk_entries = exclusive_scan_i64(sums, offs, ntiles);
            MI_LAUNCH(k_hub_scan_apply, dim3((unsigned)ntiles), dim3(256), c.stream, h->ptrb.as<int32_t>(), ntab + 1, (const int64_t*)offs);
            h->lcol.alloc(sizeof(uint16_t) * (size_t)(nblk_entries + 1));
            h->lval.alloc(sizeof(T) * (size_t)(nblk_entries + 1));
            MI_LAUNCH((k_hub_blk_fill<T>), dim3(egrid), dim3(256), c.stream, B.nnz, (const int32_t*)rowid_b.as<int32_t>(), (const int64_t*)Bs.ptr,
                      (const int32_t*)Bs.col, (const T*)Bs.val, cmax, (const uint16_t*)h->lut.as<uint16_t>(), (const int32_t*)h->bcol0.as<int32_t>(),
                      h->rows1, (const int32_t*)h->ptrb.as<int32_t>(), h->lcol.as<uint16_t>(), h->lval.as<T>());
            MI_HIP_CHECK(hipStreamSynchronize(c.stream));  // rowid_b goes (not Context::sync: that would release retired scratch arenas -- bd.ub may live in one)
            if (o.trace_phases)
                fprintf(stderr, "[mi_sparse spgemm] hub: %d blocks (<= %d columns) over %d of %lld columns, %lld of %lld entries of B; %lld hub rows, %lld items\n",
                        nbd, h->wmax, (int)cmax, (long long)cols, (long long)nblk_entries, (long long)B.nnz, (long long)h->nhub, (long long)h->n_items);
        }
        trace_mark("hub: block-major copy", t_last);
        // 6. the range path keeps what is right of every hub row's floor
        MI_LAUNCH(k_hub_cut, dim3((unsigned)std::min<int64_t>(h->nhub, (int64_t)1 << 20)), dim3(256), c.stream, h->nhub,
                  (const HubRow*)h->hrow.as<HubRow>(), (const int32_t*)A.col, (const int64_t*)Bs.ptr, (const int32_t*)Bs.col,
                  (const int32_t*)h->cfloor.as<int32_t>(), st.big.ext0.as<int64_t>(), st.big.extlen.as<int32_t>(), bd.ub);
        bd.stats = device_row_stats(bd.ub, A.rows);
        bd.sum_ub = bd.stats.sum;
        bd.max_ub = bd.stats.max;
        bd.lists_valid = false;
        h->dcnt.alloc(sizeof(int32_t) * (size_t)nbd * (size_t)h->nhub);
        h->doff.alloc(sizeof(int32_t) * (size_t)nbd * (size_t)h->nhub);
        trace_mark("hub: floors", t_last);
    } catch (const status_error& e) {
        if (e.status != MI_SPARSE_STATUS_ALLOC_FAILED) throw;
        // the extents may have been cut already: the caller recomputes its bounds
        clear_error();
        (void)hipStreamSynchronize(c.stream);
        return false;
    }
    st.big.b_sorted = true;
    st.b_gen = h->Bs.order_gen;  // the extents were laid out for the relabelled copy (spgemm_numeric would rebuild them -- whole rows -- for an operand whose entries moved)
    st.big.cfloor = h->cfloor.as<int32_t>();
    st.big.want_rank = false;
    st.upper_mode = 0;
    spgemm_symbolic<T>(A, h->Bs, C, st, bd, h.get());
    if (!st.big.have_bounds && bd.max_ub > hub_min)
        fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "hub path: the range path recorded no range starts");
    spgemm_numeric<T>(A, h->Bs, C, st, &bd, h.get());
    if (C.nnz > 0)  // the range-path part of every row (its first row_nnz entries) is in relabelled columns: map it back
        MI_LAUNCH(k_hub_unmap, dim3((unsigned)std::min<int64_t>(ceil_div(C.rows, 4), (int64_t)1 << 20)), dim3(256), c.stream, C.rows,
                  (const int64_t*)C.ptr, (const int64_t*)st.row_nnz.as<int64_t>(), (const int32_t*)h->inv.as<int32_t>(), C.col);
    c.sync();  // the relabelled copies are released on return
    st.big.cfloor = nullptr;
    counters().spgemm_hub_items += (double)h->n_items;
    return true;
}

// C := A * B (or its upper triangle) for bounds `bd` / state `st` already computed by spgemm_bounds.
template <typename T>
static void spgemm_core(const Csr& A, const Csr& B, bool upper, Csr& C, SpgemmSymbolic& st, SpgemmBounds& bd, bool want_sorted = false)
{
    st.big.want_rank = want_sorted;
    // Hub rows through dense accumulators over popularity-ordered column blocks (round 6, spgemm_hub.inc): takes B sorted or
    // not (it multiplies by a relabelled, sorted copy)
    if (!upper && !options().deterministic) {
        const int64_t ub_max0 = bd.max_ub;
        if (spgemm_hub<T>(A, B, C, st, bd)) return;
        if (bd.max_ub != ub_max0) {  // declined after the extents were cut (out of memory half way): start over
            C = Csr();
            bd = spgemm_bounds<T>(A, B, upper, C, st);
            st.big.want_rank = want_sorted;
        }
    }
    // Sort on ingest (round 5): rows too long for the LDS hash tables need B's rows in column order (the bitmap path cuts
    // them into column ranges by search); with unsorted rows they fell to the global-memory hash -- correct, and several
    // times slower.  mkl_sparse_spmm takes unsorted input without penalty (reference _sparse_sparse.py:35-40), so: a sorted
    // COPY of B for this product (B itself may alias caller memory and is never reordered behind the caller's back).
    if (!st.big.b_sorted && bd.max_ub > 4096 && options().spgemm_sort_ingest && B.nnz > 0) {
        Context& c = ctx();
        Csr Bs;
        Bs.rows = B.rows;
        Bs.cols = B.cols;
        Bs.nnz = B.nnz;
        Bs.ptr = B.ptr;  // rows keep their extents
        Bs.col_own.alloc(sizeof(int32_t) * (size_t)B.nnz);
        Bs.val_own.alloc(sizeof(T) * (size_t)B.nnz);
        Bs.col = Bs.col_own.as<int32_t>();
        Bs.val = Bs.val_own.p;
        MI_HIP_CHECK(hipMemcpyAsync(Bs.col, B.col, sizeof(int32_t) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
        MI_HIP_CHECK(hipMemcpyAsync(Bs.val, B.val, sizeof(T) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
        Bs.valid = true;
        sort_csr(type_char<T>::value, Bs);
        SpgemmSymbolic st2;
        SpgemmBounds bd2 = spgemm_bounds<T>(A, Bs, upper, C, st2);
        st2.big.want_rank = want_sorted;
        if (spgemm_onepass<T>(A, Bs, C, st2, bd2)) return;
        spgemm_symbolic<T>(A, Bs, C, st2, bd2);
        spgemm_numeric<T>(A, Bs, C, st2, &bd2);
        c.sync();  // Bs is released on return
        return;
    }
    if (spgemm_onepass<T>(A, B, C, st, bd)) return;
    spgemm_symbolic<T>(A, B, C, st, bd);
    spgemm_numeric<T>(A, B, C, st, &bd);
}

// ---- B wider than an LDS bitmap: column panels ---------------------------------------------------------------------------
// The big-row path (k_spgemm_bitmap / k_spgemm_part) holds one bit per column of B in LDS -- about 1.1 M columns.  A wider B
// used to send every row of the product beyond the LDS hash classes to the global-memory hash (17 x slower on a power-law
// product).  mkl_sparse_spmm has no such cliff (reference _sparse_sparse.py:35-40), so: B is cut into PANELS of 2^20 columns
// (its rows are sorted -- a sorted copy is made when they are not -- so a panel is a contiguous piece of every row), the
// product of A with every panel runs on the fast path with the panel's columns rebased to 0, and the rows of the result are
// the concatenation of the panels' rows (ascending panel order; inside a panel's piece the order is the fast path's).
// The symbolic phases of all panels run first (row lengths of C = their sums), then every panel's numeric phase writes
// straight into its piece of every row of C, adding the panel's first column to the indices it writes (BigRows::col_base).  Costs: A is walked once per panel and phase; B is held twice during the call.
constexpr int64_t PANEL_COLS = (int64_t)1 << 20;

__global__ void __launch_bounds__(256)
    k_panel_extent(int64_t rows, const int64_t* __restrict__ bptr, const int32_t* __restrict__ bcol, int64_t c_lo, int64_t c_hi,
                   int64_t* __restrict__ lo, int64_t* __restrict__ len)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= rows) return;
    const int64_t b = bptr[k], e = bptr[k + 1];
    auto lower = [&](int64_t from, int64_t key) {  // first position in [from, e) whose column is >= key
        int64_t l = from, h = e;
        while (l < h) {
            const int64_t m = (l + h) >> 1;
            if ((int64_t)bcol[m] < key) l = m + 1;
            else h = m;
        }
        return l;
    };
    const int64_t s = lower(b, c_lo);
    const int64_t t = lower(s, c_hi);
    lo[k] = s;
    len[k] = t - s;
}

// one wave per row of B: its entries inside the panel, columns rebased
template <typename T>
__global__ void __launch_bounds__(256)
    k_panel_fill(int64_t rows, const int64_t* __restrict__ lo, const int64_t* __restrict__ qptr, const int32_t* __restrict__ bcol,
                 const T* __restrict__ bval, int32_t col_base, int32_t* __restrict__ qcol, T* __restrict__ qval)
{
    const int64_t k = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (k >= rows) return;
    const int64_t d = qptr[k], n = qptr[k + 1] - d, s = lo[k];
    for (int64_t t = lane; t < n; t += WAVE) {
        qcol[d + t] = bcol[s + t] - col_base;
        qval[d + t] = bval[s + t];
    }
}

__global__ void __launch_bounds__(256)
    k_panel_add_len(int64_t rows, const int64_t* __restrict__ qptr, int64_t* __restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) len[i] += qptr[i + 1] - qptr[i];
}

struct SpgemmPanel {
    Csr B, Cp;  // the panel of B (columns rebased); Cp: the panel's own row pointer / entry count (no arrays)
    SpgemmSymbolic st;
    int64_t col0 = 0;  // first column of the panel in B
};

// The panels of one product: built by panels_symbolic, consumed (any number of times: the staged API) by panels_numeric.
struct PanelSet {
    std::vector<std::unique_ptr<SpgemmPanel>> panels;  // the panels that contribute entries, ascending
    int64_t np = 0;                                    // panels B was cut into
    bool upper = false;
    int64_t a_nnz = 0, b_nnz = 0;
};

// Symbolic phase of every panel of Bp (rows in column order): row pointer and entry count of C.  Throws ALLOC_FAILED when the
// panels' copies of B do not fit (the caller falls back).
template <typename T>
static void panels_symbolic(const Csr& A, const Csr& Bp, bool upper, Csr& C, PanelSet& ps)
{
    Context& c = ctx();
    auto t_last = std::chrono::steady_clock::now();
    ps.np = ceil_div(Bp.cols, PANEL_COLS);
    ps.upper = upper;
    ps.a_nnz = A.nnz;
    ps.b_nnz = Bp.nnz;
    ps.panels.clear();
    DevBuf lo_b, len_b;
    lo_b.alloc(sizeof(int64_t) * (size_t)(Bp.rows + 1));
    len_b.alloc(sizeof(int64_t) * (size_t)(std::max(Bp.rows, A.rows) + 1));
    const unsigned rgrid = (unsigned)ceil_div(Bp.rows, 256);
    for (int64_t q = 0; q < ps.np; ++q) {
        auto pn = std::make_unique<SpgemmPanel>();
        pn->col0 = q * PANEL_COLS;
        Csr& Bq = pn->B;
        Bq.rows = Bp.rows;
        Bq.cols = std::min(PANEL_COLS, Bp.cols - q * PANEL_COLS);
        Bq.ptr_own.alloc(sizeof(int64_t) * (size_t)(Bp.rows + 1));
        Bq.ptr = Bq.ptr_own.as<int64_t>();
        MI_LAUNCH(k_panel_extent, dim3(rgrid), dim3(256), c.stream, Bp.rows, (const int64_t*)Bp.ptr, (const int32_t*)Bp.col,
                  q * PANEL_COLS, q * PANEL_COLS + Bq.cols, lo_b.as<int64_t>(), len_b.as<int64_t>());
        Bq.nnz = exclusive_scan_i64(len_b.as<int64_t>(), Bq.ptr, Bp.rows);
        if (Bq.nnz == 0) continue;
        Bq.col_own.alloc(sizeof(int32_t) * (size_t)Bq.nnz);
        Bq.val_own.alloc(sizeof(T) * (size_t)Bq.nnz);
        Bq.col = Bq.col_own.as<int32_t>();
        Bq.val = Bq.val_own.p;
        MI_LAUNCH((k_panel_fill<T>), dim3((unsigned)ceil_div(Bp.rows * WAVE, 256)), dim3(256), c.stream, Bp.rows,
                  (const int64_t*)lo_b.as<int64_t>(), (const int64_t*)Bq.ptr, (const int32_t*)Bp.col, (const T*)Bp.val,
                  (int32_t)(q * PANEL_COLS), Bq.col, static_cast<T*>(Bq.val));
        Bq.valid = true;
        cache_set(Bq.sorted, true);
        Bq.order_gen = next_order_gen();
        SpgemmBounds bdq = spgemm_bounds<T>(A, Bq, upper, pn->Cp, pn->st, q * PANEL_COLS);
        spgemm_symbolic<T>(A, Bq, pn->Cp, pn->st, bdq);
        if (pn->Cp.nnz == 0) continue;
        ps.panels.push_back(std::move(pn));
    }
    trace_mark("panels: symbolic", t_last);
    int64_t* len = len_b.as<int64_t>();
    MI_HIP_CHECK(hipMemsetAsync(len, 0, sizeof(int64_t) * (size_t)(A.rows + 1), c.stream));
    const unsigned agrid = (unsigned)ceil_div(A.rows, 256);
    for (auto& pn : ps.panels) MI_LAUNCH(k_panel_add_len, dim3(agrid), dim3(256), c.stream, A.rows, (const int64_t*)pn->Cp.ptr, len);
    C.rows = A.rows;
    C.cols = Bp.cols;
    C.ptr_own.alloc(sizeof(int64_t) * (size_t)(A.rows + 1));
    C.ptr = C.ptr_own.as<int64_t>();
    C.nnz = exclusive_scan_i64(len, C.ptr, A.rows);
    C.col = nullptr;
    C.val = nullptr;
    C.valid = false;
    MI_HIP_CHECK(hipStreamSynchronize(c.stream));  // lo_b / len_b go
}

// Numeric phase of every panel, straight into its pieces of the rows of C (cursor[i] = where row i continues).  `refill`: copy
// the panels' values out of Bp again first (the staged API: B's values may have changed since the symbolic phase).
template <typename T>
static void panels_numeric(const Csr& A, const Csr& Bp, Csr& C, PanelSet& ps, bool refill)
{
    Context& c = ctx();
    auto t_last = std::chrono::steady_clock::now();
    if (A.nnz != ps.a_nnz || Bp.nnz != ps.b_nnz)
        fail(MI_SPARSE_STATUS_INVALID_VALUE, "operand structure changed since the symbolic phase (nnz %lld x %lld, was %lld x %lld)",
             (long long)A.nnz, (long long)Bp.nnz, (long long)ps.a_nnz, (long long)ps.b_nnz);
    if (!C.col_own.p || C.col_own.bytes < sizeof(int32_t) * (size_t)std::max<int64_t>(C.nnz, 1)) C.col_own.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(C.nnz, 1));
    if (!C.val_own.p || C.val_own.bytes < sizeof(T) * (size_t)std::max<int64_t>(C.nnz, 1)) C.val_own.alloc(sizeof(T) * (size_t)std::max<int64_t>(C.nnz, 1));
    C.col = C.col_own.as<int32_t>();
    C.val = C.val_own.p;
    int64_t* cursor = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(A.rows + 1)));
    MI_HIP_CHECK(hipMemcpyAsync(cursor, C.ptr, sizeof(int64_t) * (size_t)A.rows, hipMemcpyDeviceToDevice, c.stream));
    const unsigned agrid = (unsigned)ceil_div(A.rows, 256);
    int64_t* lo = nullptr;
    int64_t* len_tmp = nullptr;
    if (refill) {
        lo = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(Bp.rows + 1)));
        len_tmp = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(Bp.rows + 1)));
    }
    for (auto& up : ps.panels) {
        SpgemmPanel& pn = *up;
        if (refill) {
            MI_LAUNCH(k_panel_extent, dim3((unsigned)ceil_div(Bp.rows, 256)), dim3(256), c.stream, Bp.rows, (const int64_t*)Bp.ptr,
                      (const int32_t*)Bp.col, pn.col0, pn.col0 + pn.B.cols, lo, len_tmp);
            MI_LAUNCH((k_panel_fill<T>), dim3((unsigned)ceil_div(Bp.rows * WAVE, 256)), dim3(256), c.stream, Bp.rows, (const int64_t*)lo,
                      (const int64_t*)pn.B.ptr, (const int32_t*)Bp.col, (const T*)Bp.val, (int32_t)pn.col0, pn.B.col, static_cast<T*>(pn.B.val));
        }
        if (A.order_gen != pn.st.a_gen) {
            // A's entries moved since the symbolic phase (mi_sparse_order between the stages): the per-entry extents follow its storage order
            int64_t* ub = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(A.rows + 1)));
            launch_row_ub(A, pn.B, pn.st.big.panel_upper ? 2 : pn.st.upper_mode, ub, pn.st.big.ext0.as<int64_t>(), pn.st.big.extlen.as<int32_t>(),
                          pn.st.big.panel_upper ? (int64_t)pn.st.big.diag_shift : 0);
            pn.st.a_gen = A.order_gen;
        }
        const RowStats rs = device_row_stats(pn.st.row_nnz.as<int64_t>(), A.rows);
        pn.st.big.col_base.base = (int32_t)pn.col0;  // the numeric kernels write the panel's columns at their place in B
        run_phase<T, true>(A, pn.B, pn.st.upper_mode, pn.st.row_nnz.as<int64_t>(), rs, nullptr, cursor, C.col, static_cast<T*>(C.val),
                           pn.st.big);
        MI_LAUNCH(k_panel_add_len, dim3(agrid), dim3(256), c.stream, A.rows, (const int64_t*)pn.Cp.ptr, cursor);
    }
    trace_mark("panels: numeric", t_last);
    C.valid = true;
    C.order_gen = next_order_gen();
    C.sorted = false;
    C.range_cap = 0;  // a row is the concatenation of the panels' pieces: not one sequence of full ranges
    C.sorted_min_len = 0;
    counters().spgemm_panels += (double)ps.np;
}

// false: not done (the panels' copies of B did not fit) -- the caller takes the global-memory path
template <typename T>
static bool spgemm_panels(const Csr& A, const Csr& B, bool upper, Csr& C)
{
    Context& c = ctx();
    try {
        // rows of B in column order
        const Csr* Bp = &B;
        Csr Bs;
        if (!rows_sorted(B)) {
            Bs.rows = B.rows;
            Bs.cols = B.cols;
            Bs.nnz = B.nnz;
            Bs.ptr = B.ptr;
            Bs.col_own.alloc(sizeof(int32_t) * (size_t)B.nnz);
            Bs.val_own.alloc(sizeof(T) * (size_t)B.nnz);
            Bs.col = Bs.col_own.as<int32_t>();
            Bs.val = Bs.val_own.p;
            MI_HIP_CHECK(hipMemcpyAsync(Bs.col, B.col, sizeof(int32_t) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
            MI_HIP_CHECK(hipMemcpyAsync(Bs.val, B.val, sizeof(T) * (size_t)B.nnz, hipMemcpyDeviceToDevice, c.stream));
            Bs.valid = true;
            sort_csr(type_char<T>::value, Bs);
            Bp = &Bs;
        }
        PanelSet ps;
        panels_symbolic<T>(A, *Bp, upper, C, ps);
        panels_numeric<T>(A, *Bp, C, ps, false);
        c.sync();  // the panels are released on return
    } catch (const status_error& e) {
        if (e.status != MI_SPARSE_STATUS_ALLOC_FAILED) throw;
        clear_error();
        c.sync();
        return false;
    }
    if (options().deterministic && C.nnz > 0) spgemm_values_deterministic<T>(A, B, C, upper ? 1 : 0);
    return true;
}

// C := A * B (or its upper triangle).  C's storage is allocated here.
template <typename T>
static void spgemm_typed(const Csr& A, const Csr& B, bool upper, Csr& C, bool want_sorted)
{
    SpgemmSymbolic st;
    SpgemmBounds bd = spgemm_bounds<T>(A, B, upper, C, st);
    // rows beyond the LDS hash classes and a B too wide for the LDS bitmap: column panels (see above)
    if (bd.max_ub > 4096 && options().spgemm_col_panels && !options().spgemm_force_global && B.nnz > 0 &&
        B.nnz < ((int64_t)1 << 31) && bitmap_lds_bytes(B.cols) > (size_t)140 * 1024) {
        if (spgemm_panels<T>(A, B, upper, C)) return;
        C = Csr();
        bd = spgemm_bounds<T>(A, B, upper, C, st);
    }
    spgemm_core<T>(A, B, upper, C, st, bd, want_sorted);
}

void spgemm(char vtype, const Csr& A, const Csr& B, bool upper, Csr& C, bool want_sorted = false)
{
    by_type(vtype, [&](auto tag) { spgemm_typed<decltype(tag)>(A, B, upper, C, want_sorted); });
}

// ---- staged product (mkl_sparse_sp2m analogue) -------------------------------------------------------------
// request codes are MKL's (reference _constants.py:49-53)
constexpr int STAGE_FULL_MULT = 90, STAGE_NNZ_COUNT = 91, STAGE_FINALIZE_MULT = 92, STAGE_FULL_MULT_NO_VAL = 93,
              STAGE_FINALIZE_MULT_NO_VAL = 94;

struct Sp2mState {
    SpgemmSymbolic sym;
    std::shared_ptr<PanelSet> panels;  // a B wider than the LDS bitmap of the big-row path: the product by column panels (round 6)
    const mi_sparse_matrix* a = nullptr;  // operands the pattern was computed for (identity check only)
    const mi_sparse_matrix* b = nullptr;
    int op_a = 0, op_b = 0;
    bool upper = false;
};

static Csr& operand_csr(mi_sparse_matrix* h, int op)
{
    if (op == MI_SPARSE_OPERATION_NON_TRANSPOSE) return need_csr(h);
    if (op == MI_SPARSE_OPERATION_TRANSPOSE || (op == MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE && h->vtype != 'c' && h->vtype != 'z'))
        return need_csrT(h);
    fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "operation code %d is not supported for this value type", op);
}

// Runs the stages `request` asks for on result handle *C (created here for the stages that start a product).
static void sp2m_run(int op_a, mi_sparse_matrix* ha, int op_b, mi_sparse_matrix* hb, bool upper, int request,
                     mi_sparse_matrix_t* C)
{
    if (ha->vtype != hb->vtype) fail(MI_SPARSE_STATUS_INVALID_VALUE, "operands hold different value types");
    const bool starts = request == STAGE_FULL_MULT || request == STAGE_NNZ_COUNT || request == STAGE_FULL_MULT_NO_VAL;
    const bool finishes = request != STAGE_NNZ_COUNT;
    if (!starts && request != STAGE_FINALIZE_MULT && request != STAGE_FINALIZE_MULT_NO_VAL)
        fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad request code %d", request);
    ctx().scratch_reset();
    Csr& a = operand_csr(ha, op_a);
    Csr& b = operand_csr(hb, op_b);
    mi_sparse_matrix* r = nullptr;
    std::shared_ptr<Sp2mState> st;
    bool created = false;
    if (starts) {
        const int ib = ha->index_bytes > hb->index_bytes ? ha->index_bytes : hb->index_bytes;
        r = new_result_handle(ha->vtype, ib, a.rows, b.cols);
        created = true;
        st = std::make_shared<Sp2mState>();
        st->a = ha;
        st->b = hb;
        st->op_a = op_a;
        st->op_b = op_b;
        st->upper = upper;
        r->staged = st;
    } else {
        r = check_handle(*C);
        st = std::static_pointer_cast<Sp2mState>(r->staged);
        if (!st || !st->sym.done)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "FINALIZE requested on a handle that did not go through NNZ_COUNT");
        if (st->a != ha || st->b != hb || st->op_a != op_a || st->op_b != op_b)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "FINALIZE requested with other operands than NNZ_COUNT");
        if (a.rows != r->rows || b.cols != r->cols || a.cols != b.rows)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "operand shapes changed since NNZ_COUNT");
    }
    try {
        by_type(ha->vtype, [&](auto tag) {
            using T = decltype(tag);
            if (starts) {  // staged: always two phases -- the pattern is kept for FINALIZE / later numeric re-runs
                SpgemmBounds bd = spgemm_bounds<T>(a, b, upper, r->csr, st->sym);
                // rows beyond the LDS hash classes and a B too wide for the LDS bitmap: column panels, as the one-shot product
                // (round 6; the global-memory hash before).  The panels hold copies of B's entries, so B's rows must be in column
                // order already (an unsorted B keeps the global-memory hash here: a sorted copy could not follow set_values).
                const bool wide = bd.max_ub > 4096 && options().spgemm_col_panels && !options().spgemm_force_global && b.nnz > 0 &&
                                  b.nnz < ((int64_t)1 << 31) && bitmap_lds_bytes(b.cols) > (size_t)140 * 1024 && rows_sorted(b);
                bool done = false;
                if (wide) {
                    try {
                        auto ps = std::make_shared<PanelSet>();
                        r->csr = Csr();
                        panels_symbolic<T>(a, b, upper, r->csr, *ps);
                        st->panels = ps;
                        st->sym.done = true;
                        done = true;
                    } catch (const status_error& e) {
                        if (e.status != MI_SPARSE_STATUS_ALLOC_FAILED) throw;
                        clear_error();
                        r->csr = Csr();
                        bd = spgemm_bounds<T>(a, b, upper, r->csr, st->sym);
                    }
                }
                if (!done) spgemm_symbolic<T>(a, b, r->csr, st->sym, bd);
            }
            if (finishes) {
                if (st->panels) {
                    panels_numeric<T>(a, b, r->csr, *st->panels, !starts);  // (a later stage: B's values may have been replaced)
                    if (options().deterministic && r->csr.nnz > 0) spgemm_values_deterministic<T>(a, b, r->csr, upper ? 1 : 0);
                } else {
                    spgemm_numeric<T>(a, b, r->csr, st->sym);
                }
            }
        });
        if (finishes && !created) {
            // a numeric re-run rewrote the values of an EXISTING result in place: whatever was derived from the old values is
            // stale -- the transpose, the packed records of the dense gram, the column-partitioned SpMM plans (value copies)
            std::lock_guard<std::mutex> lk(r->mtx);
            r->csrT = Csr();
            r->csr.gram_rec.release();
            r->plan.reset_kpart();
            r->planT.reset_kpart();
        }
        ctx().sync();
    } catch (...) {
        if (created) {
            r->magic = 0;
            delete r;
        }
        throw;
    }
    *C = r;
}

template <typename T>
static int spmmd_generic(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, T* C, int64_t ldc)
{
    return guarded([&] {
        mi_sparse_matrix* ha = check_handle(A);
        mi_sparse_matrix* hb = check_handle(B);
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE) fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "spmmd supports op = 10 only");
        if (ha->vtype != type_char<T>::value || hb->vtype != type_char<T>::value)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "value type mismatch between handles and routine");
        if (layout != MI_SPARSE_LAYOUT_ROW_MAJOR && layout != MI_SPARSE_LAYOUT_COLUMN_MAJOR)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad layout code %d", layout);
        if (ha->cols != hb->rows)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "dimension mismatch: (%lld x %lld) * (%lld x %lld)",
                 (long long)ha->rows, (long long)ha->cols, (long long)hb->rows, (long long)hb->cols);
        const int64_t m = ha->rows, n = hb->cols;
        if (m == 0 || n == 0) return;
        if (!C) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output array");
        const bool row_major = layout == MI_SPARSE_LAYOUT_ROW_MAJOR;
        if (ldc < (row_major ? n : m)) fail(MI_SPARSE_STATUS_INVALID_VALUE, "ldc too small");
        Context& c = ctx();
        c.scratch_reset();
        Csr& a = need_csr(ha);
        Csr& b = need_csr(hb);
        const int64_t c_rs = row_major ? ldc : 1, c_cs = row_major ? 1 : ldc;
        const size_t extent = row_major ? (size_t)((m - 1) * ldc + n) : (size_t)((n - 1) * ldc + m);
        Staged sc;
        sc.stage_in(C, sizeof(T) * extent, false);
        if (options().spmmd_lds) {
            const unsigned grid = (unsigned)std::min<int64_t>(m, (int64_t)std::max(c.cus, 1) * 64);
            MI_LAUNCH((k_spmmd_lds<T>), dim3(grid), dim3(1024), c.stream, m, n, (const int64_t*)a.ptr, (const int32_t*)a.col,
                      (const T*)a.val, (const int64_t*)b.ptr, (const int32_t*)b.col, (const T*)b.val, static_cast<T*>(sc.dev), c_rs,
                      c_cs);
        } else {
            MI_LAUNCH((k_fill_dense<T>), dim3((unsigned)ceil_div(m * n, 256)), dim3(256), c.stream, static_cast<T*>(sc.dev), m,
                      n, c_rs, c_cs, vt<T>::zero());
            MI_LAUNCH((k_spmmd<T>), dim3((unsigned)ceil_div(m * WAVE, 256)), dim3(256), c.stream, m, (const int64_t*)a.ptr,
                      (const int32_t*)a.col, (const T*)a.val, (const int64_t*)b.ptr, (const int32_t*)b.col,
                      (const T*)b.val, static_cast<T*>(sc.dev), c_rs, c_cs);
        }
        MI_HIP_CHECK(hipGetLastError());
        sc.copy_back();
    });
}

// ---- A B A^T with symmetric B (mkl_sparse_sypr analogue) -----------------------------------------------------
// Only the UPPER triangle of B is referenced (descr: symmetric, fill mode upper).  The full symmetric matrix is
// laid out once -- row i = { B^T[i, k] : k < i } followed by { B[i, j] : j >= i } (sorted when B's rows are) --
// and the product is two SpGEMMs, the second one restricted to the upper triangle.
__global__ void k_sym_count(int64_t n, const int64_t* __restrict__ uptr, const int32_t* __restrict__ ucol,
                            const int64_t* __restrict__ tptr, const int32_t* __restrict__ tcol, int64_t* __restrict__ len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t c = 0;
    for (int64_t p = uptr[i]; p < uptr[i + 1]; ++p) c += ucol[p] >= i;
    for (int64_t p = tptr[i]; p < tptr[i + 1]; ++p) c += tcol[p] < i;
    len[i] = c;
}
template <typename T>
__global__ void k_sym_fill(int64_t n, const int64_t* __restrict__ uptr, const int32_t* __restrict__ ucol,
                           const T* __restrict__ uval, const int64_t* __restrict__ tptr, const int32_t* __restrict__ tcol,
                           const T* __restrict__ tval, const int64_t* __restrict__ optr, int32_t* __restrict__ ocol,
                           T* __restrict__ oval)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t o = optr[i];
    for (int64_t p = tptr[i]; p < tptr[i + 1]; ++p)
        if (tcol[p] < i) {
            ocol[o] = tcol[p];
            oval[o++] = tval[p];
        }
    for (int64_t p = uptr[i]; p < uptr[i + 1]; ++p)
        if (ucol[p] >= i) {
            ocol[o] = ucol[p];
            oval[o++] = uval[p];
        }
}

template <typename T>
static void symmetric_expand(const Csr& u, const Csr& ut, Csr& out)
{
    Context& c = ctx();
    const int64_t n = u.rows;
    out.rows = out.cols = n;
    out.ptr_own.alloc(sizeof(int64_t) * (size_t)(n + 1));
    out.ptr = out.ptr_own.as<int64_t>();
    int64_t* len = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(n + 1)));
    if (n)
        MI_LAUNCH(k_sym_count, dim3((unsigned)ceil_div(n, 256)), dim3(256), c.stream, n, (const int64_t*)u.ptr,
                  (const int32_t*)u.col, (const int64_t*)ut.ptr, (const int32_t*)ut.col, len);
    out.nnz = exclusive_scan_i64(len, out.ptr, n);
    out.col_own.alloc(sizeof(int32_t) * (size_t)out.nnz);
    out.val_own.alloc(sizeof(T) * (size_t)out.nnz);
    out.col = out.col_own.as<int32_t>();
    out.val = out.val_own.p;
    if (n)
        MI_LAUNCH((k_sym_fill<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), c.stream, n, (const int64_t*)u.ptr,
                  (const int32_t*)u.col, (const T*)u.val, (const int64_t*)ut.ptr, (const int32_t*)ut.col, (const T*)ut.val,
                  (const int64_t*)out.ptr, out.col, static_cast<T*>(out.val));
    out.valid = true;
    out.order_gen = next_order_gen();
    out.sorted = u.sorted && ut.sorted;
}

// dense symmetric operand given by its upper triangle -> full square, row-major, ld = n
template <typename T>
__global__ void k_sym_dense(int64_t n, const T* __restrict__ B, int64_t b_rs, int64_t b_cs, T* __restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t % n;
    out[t] = (j >= i) ? B[i * b_rs + j * b_cs] : B[j * b_rs + i * b_cs];
}
// C(upper) := alpha * P + beta * C(upper);  P full n x n row-major, ld = n
template <typename T>
__global__ void k_axpby_upper(int64_t n, const T* __restrict__ P, T alpha, T beta, int beta_zero, T* __restrict__ C,
                              int64_t c_rs, int64_t c_cs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * n) return;
    const int64_t i = t / n, j = t % n;
    if (j < i) return;
    T* c = C + i * c_rs + j * c_cs;
    *c = beta_zero ? alpha * P[t] : alpha * P[t] + beta * (*c);
}

template <typename T>
void spmm_device(mi_sparse_matrix* h, bool transposed, const Csr& m, int conj_a, T alpha, int layout, const T* B,
                 int64_t N, int64_t ldb, T beta, T* C, int64_t ldc);  // spmm.hip

template <typename T>
static int syprd_generic(int op, mi_sparse_matrix_t A, const T* B, int layout_b, int64_t ldb, T alpha, T beta, T* C,
                         int layout_c, int64_t ldc)
{
    return guarded([&] {
        mi_sparse_matrix* h = check_handle(A);
        if (h->vtype != type_char<T>::value)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "handle holds '%c' values but the '%c' routine was called", h->vtype,
                 type_char<T>::value);
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE && op != MI_SPARSE_OPERATION_TRANSPOSE)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad operation code %d", op);
        for (int l : {layout_b, layout_c})
            if (l != MI_SPARSE_LAYOUT_ROW_MAJOR && l != MI_SPARSE_LAYOUT_COLUMN_MAJOR)
                fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad layout code %d", l);
        const bool trans = op == MI_SPARSE_OPERATION_TRANSPOSE;
        const int64_t m = trans ? h->cols : h->rows, k = trans ? h->rows : h->cols;  // op(A) is m x k, B k x k, C m x m
        if (m == 0) return;
        if (!C || (!B && k > 0)) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL dense operand");
        if (ldb < k || ldc < m) fail(MI_SPARSE_STATUS_INVALID_VALUE, "leading dimension too small");
        Context& c = ctx();
        c.scratch_reset();
        Csr& a = trans ? need_csrT(h) : need_csr(h);  // CSR of op(A)
        const bool brm = layout_b == MI_SPARSE_LAYOUT_ROW_MAJOR, crm = layout_c == MI_SPARSE_LAYOUT_ROW_MAJOR;
        Staged sb, sc;
        sb.stage_in(B, sizeof(T) * (size_t)(k ? (k - 1) * ldb + k : 0), true);
        sc.stage_in(C, sizeof(T) * (size_t)((m - 1) * ldc + m), true);
        T* bs = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)(k * k + 1)));
        T* y = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)(m * k + 1)));
        T* p = static_cast<T*>(c.scratch_alloc(sizeof(T) * (size_t)(m * m + 1)));
        if (k)
            MI_LAUNCH((k_sym_dense<T>), dim3((unsigned)ceil_div(k * k, 256)), dim3(256), c.stream, k,
                      static_cast<const T*>(sb.dev), brm ? ldb : (int64_t)1, brm ? (int64_t)1 : ldb, bs);
        // Y = op(A) Bsym  (m x k, row-major);  P = op(A) Y^T: Y read as a k x m column-major operand, P written
        // column-major (= its own transpose, P is symmetric)
        spmm_device<T>(h, trans, a, 0, vt<T>::one(), MI_SPARSE_LAYOUT_ROW_MAJOR, bs, k, k, vt<T>::zero(), y, k);
        if (k == 0) MI_HIP_CHECK(hipMemsetAsync(p, 0, sizeof(T) * (size_t)(m * m), c.stream));
        else spmm_device<T>(h, trans, a, 0, vt<T>::one(), MI_SPARSE_LAYOUT_COLUMN_MAJOR, y, m, k, vt<T>::zero(), p, m);
        MI_LAUNCH((k_axpby_upper<T>), dim3((unsigned)ceil_div(m * m, 256)), dim3(256), c.stream, m, (const T*)p, alpha, beta,
                  (int)(vt<T>::is_zero(beta) ? 1 : 0), static_cast<T*>(sc.dev), crm ? ldc : (int64_t)1,
                  crm ? (int64_t)1 : ldc);
        MI_HIP_CHECK(hipGetLastError());
        if (sb.host) c.sync();
        sc.copy_back();
    });
}

template <typename T>
static int set_values_generic(mi_sparse_matrix_t A, const T* values)
{
    return guarded([&] {
        mi_sparse_matrix* h = check_handle(A);
        if (h->vtype != type_char<T>::value)
            fail(MI_SPARSE_STATUS_INVALID_VALUE, "handle holds '%c' values but the '%c' routine was called", h->vtype,
                 type_char<T>::value);
        if (!values) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL value array");
        Context& c = ctx();
        c.scratch_reset();
        const bool created_csc = (h->origin == 'c');
        Csr& primary = created_csc ? need_csrT(h) : need_csr(h);
        if (h->origin == 'b') fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "set_values on a handle created from BSR arrays");
        if (!primary.val_own.p) {  // values were aliased from caller memory: take ownership of a copy first
            primary.val_own.alloc(sizeof(T) * (size_t)primary.nnz);
            primary.val = primary.val_own.p;
        }
        if (primary.nnz) {
            if (locate(values) == Loc::Device)
                MI_HIP_CHECK(hipMemcpyAsync(primary.val, values, sizeof(T) * (size_t)primary.nnz, hipMemcpyDeviceToDevice, c.stream));
            else
                copy_h2d(primary.val, values, sizeof(T) * (size_t)primary.nnz);
        }
        // the derived representation and the packed records of the dense gram (nothing else: plans depend on the pattern only) are stale
        primary.gram_rec.release();
        h->plan.reset_kpart();  // the column-partitioned SpMM plans hold copies of the values
        h->planT.reset_kpart();
        Csr& other = created_csc ? h->csr : h->csrT;
        other = Csr();
        c.sync();
    });
}

}  // namespace mi

using mi::cdouble;
using mi::cfloat;

extern "C" {

static mi_sparse_status_t spmm_entry(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, mi_sparse_matrix_t* C, bool ordered)
{
    return mi::guarded([&] {
        if (!C) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output handle pointer");
        *C = nullptr;
        mi_sparse_matrix* ha = mi::check_handle(A);
        mi_sparse_matrix* hb = mi::check_handle(B);
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE)
            mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "spmm supports op = 10 only");
        if (ha->vtype != hb->vtype) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "operands hold different value types");
        if (ha->cols != hb->rows)
            mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "dimension mismatch: (%lld x %lld) * (%lld x %lld)",
                     (long long)ha->rows, (long long)ha->cols, (long long)hb->rows, (long long)hb->cols);
        mi::ctx().scratch_reset();
        mi::Csr& a = mi::need_csr(ha);
        mi::Csr& b = mi::need_csr(hb);
        const int ib = ha->index_bytes > hb->index_bytes ? ha->index_bytes : hb->index_bytes;
        mi_sparse_matrix* r = mi::new_result_handle(ha->vtype, ib, ha->rows, hb->cols);
        // MKL keeps the operands' format: the product of two BSR handles with one block size is exportable as BSR
        if (ha->bsr.valid && hb->bsr.valid && ha->bsr.bs == hb->bsr.bs) r->result_bs = ha->bsr.bs;
        try {
            mi::spgemm(ha->vtype, a, b, false, r->csr, ordered);
            if (ordered) {
                mi::ctx().scratch_reset();
                mi::sort_csr(ha->vtype, r->csr);  // (looks at the rows first: a result that is in order already is left alone)
            }
            mi::ctx().sync();
        } catch (...) {
            r->magic = 0;
            delete r;
            throw;
        }
        *C = r;
    });
}

mi_sparse_status_t mi_sparse_spmm(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, mi_sparse_matrix_t* C)
{
    return spmm_entry(op, A, B, C, false);
}

mi_sparse_status_t mi_sparse_spmm_ordered(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, mi_sparse_matrix_t* C)
{
    return spmm_entry(op, A, B, C, true);
}

mi_sparse_status_t mi_sparse_syrk(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t* C)
{
    return mi::guarded([&] {
        if (!C) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output handle pointer");
        *C = nullptr;
        mi_sparse_matrix* h = mi::check_handle(A);
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE && op != MI_SPARSE_OPERATION_TRANSPOSE &&
            op != MI_SPARSE_OPERATION_CONJUGATE_TRANSPOSE)
            mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad operation code %d", op);
        if (h->vtype == 'c' || h->vtype == 'z')
            mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "syrk supports real values only (the reference rejects complex gram)");
        mi::ctx().scratch_reset();
        mi::Csr& a = mi::need_csr(h);
        mi::Csr& at = mi::need_csrT(h);
        const bool aat = (op == MI_SPARSE_OPERATION_NON_TRANSPOSE);
        const int64_t n = aat ? h->rows : h->cols;
        mi_sparse_matrix* r = mi::new_result_handle(h->vtype, h->index_bytes, n, n);
        try {
            if (aat) mi::spgemm(h->vtype, a, at, true, r->csr);
            else mi::spgemm(h->vtype, at, a, true, r->csr);
            mi::ctx().sync();
        } catch (...) {
            r->magic = 0;
            delete r;
            throw;
        }
        *C = r;
    });
}

mi_sparse_status_t mi_sparse_s_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, float* C, int64_t ldc)
{
    return mi::spmmd_generic<float>(op, A, B, layout, C, ldc);
}
mi_sparse_status_t mi_sparse_d_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, double* C, int64_t ldc)
{
    return mi::spmmd_generic<double>(op, A, B, layout, C, ldc);
}
mi_sparse_status_t mi_sparse_c_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, mi_complex8* C,
                                     int64_t ldc)
{
    return mi::spmmd_generic<cfloat>(op, A, B, layout, (cfloat*)C, ldc);
}
mi_sparse_status_t mi_sparse_z_spmmd(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, int layout, mi_complex16* C,
                                     int64_t ldc)
{
    return mi::spmmd_generic<cdouble>(op, A, B, layout, (cdouble*)C, ldc);
}

mi_sparse_status_t mi_sparse_sp2m(int op_a, struct mi_matrix_descr descr_a, mi_sparse_matrix_t A, int op_b,
                                  struct mi_matrix_descr descr_b, mi_sparse_matrix_t B, int request, mi_sparse_matrix_t* C)
{
    return mi::guarded([&] {
        if (!C) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output handle pointer");
        if (descr_a.type != MI_SPARSE_MATRIX_TYPE_GENERAL || descr_b.type != MI_SPARSE_MATRIX_TYPE_GENERAL)
            mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "sp2m supports SPARSE_MATRIX_TYPE_GENERAL descriptors only");
        mi::sp2m_run(op_a, mi::check_handle(A), op_b, mi::check_handle(B), false, request, C);
    });
}

mi_sparse_status_t mi_sparse_sypr(int op, mi_sparse_matrix_t A, mi_sparse_matrix_t B, struct mi_matrix_descr descr_b,
                                  mi_sparse_matrix_t* C, int request)
{
    return mi::guarded([&] {
        if (!C) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output handle pointer");
        *C = nullptr;
        mi_sparse_matrix* ha = mi::check_handle(A);
        mi_sparse_matrix* hb = mi::check_handle(B);
        if (request != mi::STAGE_FULL_MULT) mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "sypr supports SPARSE_STAGE_FULL_MULT only");
        if (descr_b.type != MI_SPARSE_MATRIX_TYPE_SYMMETRIC || descr_b.mode != MI_SPARSE_FILL_MODE_UPPER)
            mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "sypr needs B described as symmetric with the upper triangle stored");
        if (op != MI_SPARSE_OPERATION_NON_TRANSPOSE && op != MI_SPARSE_OPERATION_TRANSPOSE)
            mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad operation code %d", op);
        if (ha->vtype != hb->vtype) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "operands hold different value types");
        if (ha->vtype == 'c' || ha->vtype == 'z') mi::fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "sypr supports real values only");
        const bool trans = op == MI_SPARSE_OPERATION_TRANSPOSE;
        if (hb->rows != hb->cols || (trans ? ha->rows : ha->cols) != hb->rows)
            mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "dimension mismatch: op(A) is %lld x %lld, B is %lld x %lld",
                     (long long)(trans ? ha->cols : ha->rows), (long long)(trans ? ha->rows : ha->cols),
                     (long long)hb->rows, (long long)hb->cols);
        mi::ctx().scratch_reset();
        mi::Csr& x = trans ? mi::need_csrT(ha) : mi::need_csr(ha);   // op(A)
        mi::Csr& xt = trans ? mi::need_csr(ha) : mi::need_csrT(ha);  // op(A)^T
        mi::Csr& u = mi::need_csr(hb);
        mi::Csr& ut = mi::need_csrT(hb);
        const int64_t m = x.rows;
        mi_sparse_matrix* r = mi::new_result_handle(ha->vtype, ha->index_bytes, m, m);
        try {
            mi::Csr full, t;
            mi::by_type(ha->vtype, [&](auto tag) {
                using T = decltype(tag);
                if constexpr (!mi::vt<T>::is_complex) mi::symmetric_expand<T>(u, ut, full);
            });
            mi::spgemm(ha->vtype, x, full, false, t);        // T = op(A) Bsym
            mi::spgemm(ha->vtype, t, xt, true, r->csr);      // C = triu(T op(A)^T)
            mi::ctx().sync();
        } catch (...) {
            r->magic = 0;
            delete r;
            throw;
        }
        *C = r;
    });
}

mi_sparse_status_t mi_sparse_s_syprd(int op, mi_sparse_matrix_t A, const float* B, int layout_b, int64_t ldb, float alpha,
                                     float beta, float* C, int layout_c, int64_t ldc)
{
    return mi::syprd_generic<float>(op, A, B, layout_b, ldb, alpha, beta, C, layout_c, ldc);
}
mi_sparse_status_t mi_sparse_d_syprd(int op, mi_sparse_matrix_t A, const double* B, int layout_b, int64_t ldb, double alpha,
                                     double beta, double* C, int layout_c, int64_t ldc)
{
    return mi::syprd_generic<double>(op, A, B, layout_b, ldb, alpha, beta, C, layout_c, ldc);
}

mi_sparse_status_t mi_sparse_s_set_values(mi_sparse_matrix_t A, const float* values) { return mi::set_values_generic<float>(A, values); }
mi_sparse_status_t mi_sparse_d_set_values(mi_sparse_matrix_t A, const double* values) { return mi::set_values_generic<double>(A, values); }
mi_sparse_status_t mi_sparse_c_set_values(mi_sparse_matrix_t A, const mi_complex8* values)
{
    return mi::set_values_generic<cfloat>(A, (const cfloat*)values);
}
mi_sparse_status_t mi_sparse_z_set_values(mi_sparse_matrix_t A, const mi_complex16* values)
{
    return mi::set_values_generic<cdouble>(A, (const cdouble*)values);
}

}  // extern "C"
