// gram.hip -- dense-output gram product of a sparse matrix.
// Replaces mkl_sparse_?_syrkd (reference sparse_dot_mkl/_gram_matrix.py:149-157):
//     op = 10 :  C := alpha * A   * A^T + beta * C      (n = rows of A)
//     op = 11 :  C := alpha * A^T * A   + beta * C      (n = cols of A)
// Only the upper triangle (col >= row) of C is read or written.
//
// Row-owned formulation (no inter-workgroup races): with X = A (op 11) or X = A^T (op 10),
// C = X^T X and output row i is  sum over nonzeros (r, i) of X of  X[r, i] * X[r, i:].
// The CSR of X^T (cached on the handle) lists those nonzeros; a workgroup owns (row i, a 64 KiB
// column tile), its waves walk those nonzeros and let their lanes span the entries of X's row r.
#include "common.hpp"

namespace mi {

// One workgroup per (output row i, column tile): the tile of the dense row lives in LDS, products are
// accumulated with LDS float / double atomics (entries of different source rows r collide on the
// same column), and the finished tile is written to HBM exactly once, coalesced, with beta applied
// on the way -- no global atomics, no separate scaling pass.  Tiles start at the diagonal (only
// col >= row is produced).
// Finished tile -> C (defined below).  Row-major with beta = 0 (the reference's call): 16-byte NON-TEMPORAL stores -- the
// tile is never read again by this kernel -- and the tile is cleared on the way (every element a thread reads it zeroes).
template <typename T>
__device__ __forceinline__ void syrkd_flush_tile(T* acc, T* crow, int64_t c_cs, int64_t j_lo, int64_t j_hi, int64_t tile_lo,
                                                 T beta, int beta_zero, int tid, int nthreads);

// LDS tile of one workgroup: TKB KiB of accumulators.  64 KiB (two 512-thread workgroups per CU) or 128 KiB (one
// 1024-thread workgroup per CU): a wider tile halves the number of passes over the row's nonzeros when the output row
// is wider than one tile (every pass re-reads all of them and keeps ~tile / n of the entries).
template <typename T, int TKB>
constexpr int syrkd_tile() { return (int)(TKB * 1024 / sizeof(T)); }

template <typename T, int TKB>
__global__ void __launch_bounds__(TKB == 64 ? 512 : 1024)
    k_syrkd_lds(int64_t n, int64_t row0, int64_t row_end, int64_t tiles_per_row, const int64_t* __restrict__ tptr, const int32_t* __restrict__ tcol,
                const T* __restrict__ tval, const int64_t* __restrict__ xptr, const int32_t* __restrict__ xcol,
                const T* __restrict__ xval, T* __restrict__ C, int64_t c_rs, int64_t c_cs, T alpha, T beta,
                int beta_zero, int64_t n_virtual)
{
    constexpr int TILE = syrkd_tile<T, TKB>();
    __shared__ __attribute__((aligned(16))) T acc[TILE];
    for (int k = threadIdx.x; k < TILE; k += blockDim.x) acc[k] = vt<T>::zero();  // once: every flush leaves the tile zero again
    __syncthreads();
    // PERSISTENT workgroups: the grid is a few workgroups per CU and each walks the (row, tile) list with stride
    // gridDim.x (a multiple of 8, so the XCD of a list position stays position % 8).  A tile workgroup owns the whole
    // LDS of its CU; launched one per tile, every tile paid a dispatch + wave launch with the CU idle in between,
    // and the empty tiles below the diagonal paid it too.
    for (int64_t vb = blockIdx.x; vb < n_virtual; vb += gridDim.x) {
    // XCD-affine order: workgroup b runs on XCD b % 8 (observed; speed only).  All column tiles of one output row go
    // to the SAME XCD, back to back, so the rows of X that the row's nonzeros select (the same for every tile) are
    // fetched from HBM once and then served by that XCD's L2 -- with one tile per XCD in round-robin order every
    // tile missed (rocprof: FETCH = 100 x the matrix at the literal configs[3]).
    const int64_t q = vb >> 3;
    const int64_t i = row0 + (q / tiles_per_row) * 8 + (vb & 7);  // output row (C points at row `row0`)
    const int64_t t = q % tiles_per_row;
    const int64_t j_lo = i + t * TILE;
    if (i >= row_end || j_lo >= n) continue;  // uniform for the whole workgroup
    const int64_t j_hi = (j_lo + TILE < n) ? j_lo + TILE : n;
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid / WAVE, lane = tid % WAVE, nwaves = nthreads / WAVE;
    // Each wave takes 64 nonzeros (r, X[r,i]) of column i at a time: lane l fetches entry l and the
    // extent of X's row r (coalesced + one gather), then the wave walks those 64 rows with the
    // per-row scalars broadcast by readlane -- one dependent memory latency per row instead of
    // three, and four rows' loads in flight together.
    const int64_t t0 = tptr[i], t1 = tptr[i + 1];
    for (int64_t base = t0 + (int64_t)wave * WAVE; base < t1; base += (int64_t)nwaves * WAVE) {
        // every load below is UNCONDITIONAL (lanes / rows past the end read a harmless valid element and are masked
        // afterwards): a predicated load is a branch around the load, and behind branches the compiler's wait-count
        // bookkeeping falls back to vmcnt(0) -- which silently serialised the two-step pipeline further down
        const int64_t p = base + lane;
        const bool valid = p < t1;
        const int64_t p_safe = valid ? p : base;  // base < t1
        const int32_t r = tcol[p_safe];
        const T a = valid ? vt<T>::mul(alpha, tval[p_safe]) : vt<T>::zero();
        const int64_t q0 = xptr[r];
        const int64_t q1 = valid ? xptr[r + 1] : q0;
        const int cnt = (t1 - base < WAVE) ? (int)(t1 - base) : WAVE;
        // R rows per step, two steps in flight: the loads of step s + 1 are issued before the LDS atomics of step s,
        // so the walk is a pipeline of independent loads instead of one dependent round trip per step
        constexpr int R = 4;
        struct Step {
            int64_t qs[R], qe[R];
            T ae[R], xv[R];
            int32_t jj[R];
        };
        auto issue = [&](int e0, Step& st) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int e = (e0 + u < cnt) ? e0 + u : (e0 < cnt ? e0 : 0);  // clamp: duplicates are masked out below
                st.qs[u] = lane_bcast(q0, e);
                st.qe[u] = (e0 + u < cnt) ? lane_bcast(q1, e) : st.qs[u];
                st.ae[u] = lane_bcast(a, e);
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {  // first 64 entries of each row: loads issued together
                const int64_t q = st.qs[u] + lane;
                const bool ok = q < st.qe[u];
                const int64_t q_safe = ok ? q : t0;  // X^T and X hold the same number of entries: t0 indexes both
                const int32_t jl = xcol[q_safe];
                st.xv[u] = xval[q_safe];
                st.jj[u] = ok ? jl : -1;
            }
        };
        auto consume = [&](const Step& st) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int64_t j = st.jj[u];
                if (j >= j_lo && j < j_hi) lds_accum(&acc[j - j_lo], vt<T>::mul(st.ae[u], st.xv[u]));
            }
#pragma unroll
            for (int u = 0; u < R; ++u) {  // rows longer than one wave
                for (int64_t q = st.qs[u] + WAVE + lane; q < st.qe[u]; q += WAVE) {
                    const int64_t j = xcol[q];
                    if (j >= j_lo && j < j_hi) lds_accum(&acc[j - j_lo], vt<T>::mul(st.ae[u], xval[q]));
                }
            }
        };
        Step s0, s1;
        issue(0, s0);
        for (int e0 = 0; e0 < cnt; e0 += 2 * R) {
            issue(e0 + R, s1);       // past the end: every row is clamped to an empty extent
            consume(s0);
            issue(e0 + 2 * R, s0);
            consume(s1);
        }
    }
    __syncthreads();
    syrkd_flush_tile(acc, C + (i - row0) * c_rs, c_cs, j_lo, j_hi, j_lo, beta, beta_zero, tid, nthreads);  // out, and zero again
    __syncthreads();  // the tile is reused by the next list position
    }
}

// ---- sliced variant (rows of X sorted): the default ------------------------------------------------------------
// The tiles lie on a GLOBAL grid (tile g = columns [g * TILE, (g + 1) * TILE); the tile holding the diagonal is cut at
// column i).  A wave takes 64 selected rows at a time (lane = row), then EIGHT LANES walk each row's slice, eight rows
// per step, all steps' loads issued before the first LDS atomic.
//
// What the kernel is made of is L2 / HBM LINE REQUESTS, not bytes (profiles/r03_fetch_size_calibration.log: a gather
// from beyond the caches costs a 128-byte line fill however few bytes it uses; counters of round 2's kernel at the
// literal configs[3]: 4.8 k L2 read requests per tile -- per selected row one line each for its row pointer, its
// slice offsets, its column indices and its values -- 2.2e9 of them reaching HBM = 284 GB for a 2 GB matrix).  Round 3
// lays the data out so that a (row, tile) pair costs TWO lines instead of four:
//   * `off[r][g]` holds ABSOLUTE positions (row start + entries left of tile boundary g): no load of the row pointer;
//   * the entries of X are read from a packed copy of (column, value) RECORDS (SpEntry<T>, 8 bytes for float): a slice
//     of 8 entries is one 64-byte run in one array instead of 32 + 32 bytes in two.
// Both are built once per handle (k_gram_offsets, k_gram_pack) and cached on it.
// The tile loop prefetches the dependent chain  row pointer of X^T -> (r, X[r,i]) -> slice offsets  three tiles deep
// (one level per tile, all loads of a tile's burst issued together after the barrier that ends the accumulation), and
// the write-out re-zeroes the tile as it reads it.  Requesting the slice ENTRIES a tile ahead as well was measured
// 17 % slower (124 vs 106 ms, profiles/r03_gram_variants.log) and is not done.
__global__ void k_gram_offsets(int64_t rows, int64_t G, int64_t w, const int64_t* __restrict__ xptr,
                               const int32_t* __restrict__ xcol, int32_t* __restrict__ off)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * (G + 1)) return;
    const int64_t r = t / (G + 1), g = t - r * (G + 1);
    int64_t lo = xptr[r], hi = xptr[r + 1];
    const int64_t key = g * w;  // first position with column >= key (g == G: past every column)
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)xcol[mid] < key) lo = mid + 1; else hi = mid;
    }
    off[t] = (int32_t)lo;  // absolute (the sliced walk is only chosen for nnz < 2^31)
}

template <typename T>
__global__ void k_gram_pack(int64_t nnz, const int32_t* __restrict__ col, const T* __restrict__ val, SpEntry<T>* __restrict__ rec)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x) {
        SpEntry<T> e;
        e.c = col[k];
        e.v = val[k];
        rec[k] = e;
    }
}

// Slice bounds carried by the ENTRIES OF X^T (rows of X with at most 255 entries, at most 11 tiles per output row): for
// every entry (r, i) of X^T a 16-byte record { start of row r in `rec`, entries of row r left of tile boundary 0 .. G }.
// A tile then reads its selected rows' bounds as a COALESCED stream next to (r, X[r,i]) -- 16 bytes per selected row --
// instead of one random 128-byte line of the per-row table each: the table line was one of the ~2.5 lines a (row, tile)
// pair cost, and with the bounds in hand one level of the dependent chain disappears.  Built once per handle and tile
// width from the per-row table (k_gram_heads); 16 bytes per nonzero.
struct alignas(16) GramHead {
    int32_t start;
    uint32_t w0, w1, w2;  // byte g of (w0, w1, w2) = entries of the row left of tile boundary g
    __host__ __device__ __forceinline__ int left(int g) const
    {
        // g is uniform.  All three words are USED (shifted), the select is between computed values: as `g < 4 ? w[0] : g < 8 ? w[1] : w[2]` over an
        // array the compiler folded the selects of loads into one load at a selected address, which kept the 16-byte
        // record in memory -- an alloca it then promoted to LDS (32 KB next to the 128 KB tile): every tile paid a
        // ds_write_b128 + three ds_read_b32 per lane and waited for the record it had just requested.
        const uint64_t lo = ((uint64_t)w1 << 32) | w0;
        const uint32_t a = (uint32_t)(lo >> ((g & 7) * 8)), b = w2 >> ((g & 3) * 8);  // boundaries 0-7 / 8-11
        return (int)((g < 8 ? a : b) & 255u);
    }
};
constexpr int GRAM_HEAD_MAXG = 11;

__global__ void k_gram_heads(int64_t nnz, int64_t G, const int32_t* __restrict__ tcol, const int32_t* __restrict__ off,
                             GramHead* __restrict__ head)
{
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x) {
        const int32_t* orow = off + (int64_t)tcol[q] * (G + 1);
        GramHead h;
        h.start = orow[0];
        uint32_t w[3] = {0u, 0u, 0u};
#pragma unroll
        for (int g = 0; g < 12; ++g) w[g >> 2] |= (uint32_t)((orow[g <= G ? g : G] - h.start) & 255) << ((g & 3) * 8);
        h.w0 = w[0];
        h.w1 = w[1];
        h.w2 = w[2];
        head[q] = h;
    }
}

__global__ void k_gram_max_row(int64_t rows, const int64_t* __restrict__ ptr, long long* __restrict__ out)
{
    long long m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
        const long long l = (long long)(ptr[i + 1] - ptr[i]);
        if (l > m) m = l;
    }
    if (m) atomicMax(out, m);
}

// Finished tile -> C, and the tile back to zero (every element a thread reads it also clears).
template <typename T>
__device__ __forceinline__ void syrkd_flush_tile(T* acc, T* crow, int64_t c_cs, int64_t j_lo, int64_t j_hi, int64_t tile_lo,
                                                 T beta, int beta_zero, int tid, int nthreads)
{
    constexpr int V = 16 / (int)sizeof(T);
    if (beta_zero && c_cs == 1) {
        T* p0 = crow + j_lo;
        int64_t head = (int64_t)(((16 - (reinterpret_cast<uintptr_t>(p0) & 15)) & 15) / sizeof(T));
        if (head > j_hi - j_lo) head = j_hi - j_lo;
        const int64_t body = (j_hi - j_lo - head) / V;  // 16-byte vectors
        if (tid < head) {
            nt_store(p0 + tid, acc[j_lo - tile_lo + tid]);
            acc[j_lo - tile_lo + tid] = vt<T>::zero();
        }
        T* a0 = acc + (j_lo - tile_lo + head);
        if (((j_lo - tile_lo + head) % V) == 0) {
            // the usual case (every tile right of the diagonal one): the LDS side is 16-byte aligned as well -- one
            // ds_read_b128 + one ds_write_b128 per 16 bytes instead of four 4-byte reads and four writes (the tile is
            // declared 16-byte aligned; LDS instructions were 22 % of the kernel's wave cycles)
            vec<T, V>* av = reinterpret_cast<vec<T, V>*>(a0);
            vec<T, V> z;
#pragma unroll
            for (int v = 0; v < V; ++v) z.v[v] = vt<T>::zero();
            for (int64_t k = tid; k < body; k += nthreads) {
                const vec<T, V> x = av[k];
                nt_store16(p0 + head + k * V, x.v);
                av[k] = z;
            }
        } else {
            for (int64_t k = tid; k < body; k += nthreads) {
                nt_store16(p0 + head + k * V, a0 + k * V);
#pragma unroll
                for (int v = 0; v < V; ++v) a0[k * V + v] = vt<T>::zero();
            }
        }
        const int64_t done = head + body * V;
        if (tid < j_hi - j_lo - done) {
            nt_store(p0 + done + tid, acc[j_lo - tile_lo + done + tid]);
            acc[j_lo - tile_lo + done + tid] = vt<T>::zero();
        }
        return;
    }
    for (int64_t j = j_lo + tid; j < j_hi; j += nthreads) {
        T* c = crow + j * c_cs;
        const T v = acc[j - tile_lo];
        acc[j - tile_lo] = vt<T>::zero();
        *c = beta_zero ? v : vt<T>::fma(beta, *c, v);
    }
}

// MODE 0: one tile per row (the slice is the row); 1: slice bounds from the per-row table `off`; 2: from the records that
// travel with the entries of X^T (`head`)
template <typename T, int TKB, int MODE>
__global__ void __launch_bounds__(TKB == 64 ? 512 : 1024)
    k_syrkd_sliced(int64_t n, int64_t row0, int64_t row_end, int64_t G, const int64_t* __restrict__ tptr,
                   const int32_t* __restrict__ tcol, const T* __restrict__ tval, const int64_t* __restrict__ xptr,
                   const SpEntry<T>* __restrict__ rec, const int32_t* __restrict__ off, const GramHead* __restrict__ head,
                   T* __restrict__ C, int64_t c_rs, int64_t c_cs, T alpha, T beta, int beta_zero, int64_t n_virtual,
                   unsigned long long* __restrict__ queue)
{
    constexpr int TILE = syrkd_tile<T, TKB>();
    constexpr int SUB = 8;            // lanes per selected row
    constexpr int RPS = WAVE / SUB;   // rows per step
    constexpr int NSTEP = WAVE / RPS; // steps per 64 rows, all in flight together
    constexpr int HH = sizeof(T) <= 4 ? 2 : 1;  // halves of SUB entries requested together (registers: 8-byte values get one)
    __shared__ __attribute__((aligned(16))) T acc[TILE];
    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int wave = tid / WAVE, lane = tid % WAVE, nwaves = nthreads / WAVE;
    const int sub = lane % SUB, grp = lane / SUB;
    // list position -> (output row, tile); the positions left of the diagonal / past the row block hold nothing.
    // Workgroup b runs on XCD b % 8 (observed; speed only): all tiles of one output row go to the SAME XCD back to back, so
    // the lines of X that the row's nonzeros select are fetched from HBM once and then served by that XCD's L2.
    struct Pos {
        int64_t i, g, vb;
        bool ok;
    };
    auto decode = [&](int64_t vb, int64_t& i, int64_t& g) {
        const int64_t q = vb >> 3;
        i = row0 + (q / G) * 8 + (vb & 7);
        g = q % G;
        return i < row_end && g * TILE + TILE > i;
    };
    auto seek = [&](int64_t vb) {
        Pos p;
        p.i = row0;
        p.g = 0;
        while (vb < n_virtual && !decode(vb, p.i, p.g)) vb += gridDim.x;
        p.vb = vb;
        p.ok = vb < n_virtual;
        if (!p.ok) {  // a harmless, valid position: every load below stays unconditional
            p.i = row0;
            p.g = 0;
        }
        return p;
    };
    auto after = [&](const Pos& p) { return p.ok ? seek(p.vb + gridDim.x) : p; };
    // `queue` != nullptr (round 4): the (row, tile) pairs are PULLED, in order, from one counter per XCD instead of being walked
    // with a fixed stride.  With the stride every workgroup had its own sequence and the workgroups drifted apart over
    // thousands of tiles, so the tiles of one output row -- which read neighbouring slices of the same rows of X, sharing
    // the lines in between -- ran tens of microseconds apart, long after the XCD's L2 had dropped those lines.  Pulled from
    // a counter, the pairs of an XCD start in list order: the tiles of a row within a tile's duration of each other.
    // The counter enumerates VALID pairs only (tiles at or right of the diagonal), block of TILE rows by block.
    __shared__ unsigned long long s_pull[5];
    const int xq = (int)(blockIdx.x & 7u);
    auto pos_of = [&](unsigned long long d) {  // d-th valid pair of the rows  row0 + xq + 8 m
        Pos p;
        p.vb = 0;
        p.ok = false;
        p.i = row0;
        p.g = 0;
        const int64_t first = row0 + xq;
        for (int64_t b = row0 / TILE; b < G; ++b) {
            const int64_t lo = b * TILE > first ? b * TILE : first, hi = (b + 1) * TILE < row_end ? (b + 1) * TILE : row_end;
            if (hi <= lo) continue;
            const int64_t m_lo = (lo - first + 7) >> 3, m_hi = (hi - first + 7) >> 3;  // rows first + 8 m in [lo, hi)
            const unsigned long long w = (unsigned long long)(G - b), cnt = (unsigned long long)(m_hi - m_lo) * w;
            if (d < cnt) {
                p.i = first + 8 * (m_lo + (int64_t)(d / w));
                p.g = b + (int64_t)(d % w);
                p.ok = true;
                break;
            }
            d -= cnt;
        }
        return p;
    };
    // The dependent chain of a tile, one LEVEL per pipeline stage (this wave's first 64 selected rows; lane = row):
    //   A: extent of column i in X^T (uniform)   ->   B: entry (r, X[r,i]) [+ its slice bounds, MODE 2]   ->
    //   C: the slice of row r in this tile (MODE 0 / 1: one more gather)   ->   E: the slice's entries (walk_issue).
    // Every level is requested exactly ONE tile before the level that consumes it, and a level's registers hold the RAW loaded
    // data until then: nothing is computed from a load, and no loaded register is copied, in the iteration that issued it
    // (round 3 kept decoded values of several tiles in flight and rotated them: the decode / the copies made the compiler
    // wait for the loads it had just issued -- a full memory round trip per tile in front of the barrier).
    using OT = typename std::conditional<MODE == 0, int64_t, int32_t>::type;
    struct LevelA {
        int64_t t0, t1;
    };
    struct LevelB {
        T a_raw;
        int32_t start;     // MODE 2: GramHead::start ...
        uint16_t bounds;   // ... and bytes g, g + 1 of its boundary array (entries of the row left of this tile / the next one)
        int32_t r;         // MODE 0 / 1
        bool valid;
    };
    struct LevelC {
        OT o0, o1;  // the slice: absolute positions in `rec` (MODE 0 / 1: raw loads; MODE 2: decoded from B's record)
        T a;
        bool valid;
    };
    auto issue_a = [&](const Pos& p) {
        LevelA A;
        A.t0 = tptr[p.i];
        A.t1 = tptr[p.i + (p.ok ? 1 : 0)];  // an empty extent for positions that hold nothing (no branch around a load)
        return A;
    };
    auto issue_b = [&](const LevelA& A, int64_t g) {
        LevelB b;
        const int64_t base = A.t0 + (int64_t)wave * WAVE;
        int64_t q = base + lane < A.t1 ? base + lane : A.t1 - 1;  // always a valid entry ...
        if (q < 0) q = 0;                                          // ... (the kernel is only launched with nnz > 0)
        b.valid = base + lane < A.t1;
        b.a_raw = tval[q];
        b.r = 0;
        b.start = 0;
        b.bounds = 0;
        if constexpr (MODE == 2) {
            // the bounds arrive with the entry, no further level.  Of the 16-byte record only `start` and the two boundary bytes
            // of THIS tile are read (g is known when the level is requested): three single-register loads -- the 16-byte
            // tuple of a whole record was split by the register allocator while the load was in flight (a copy = a wait)
            const char* rp = reinterpret_cast<const char*>(head + q);
            b.start = *reinterpret_cast<const int32_t*>(rp);
            uint16_t two;
            __builtin_memcpy(&two, rp + 4 + g, 2);
            b.bounds = two;
        } else {
            b.r = tcol[q];
        }
        return b;
    };
    auto issue_c = [&](const LevelB& b, int64_t g) {
        LevelC c;
        c.a = vt<T>::mul(alpha, b.a_raw);
        c.valid = b.valid;
        // (a run-time branch on `off` around these loads made the compiler drain vmcnt to 0 at the join: a template flag)
        if constexpr (MODE == 2) {
            c.o0 = b.start + (int32_t)(b.bounds & 255u);
            c.o1 = b.start + (int32_t)(b.bounds >> 8);
        } else if constexpr (MODE == 1) {
            const int32_t* orow = off + (int64_t)b.r * (G + 1) + g;
            c.o0 = orow[0];
            c.o1 = orow[1];
        } else {
            c.o0 = xptr[b.r];
            c.o1 = xptr[b.r + 1];
        }
        return c;
    };
    // One tile's share of 64 selected rows, in two halves: walk_issue REQUESTS entries sub (and sub + SUB) of every row's slice
    // for all NSTEP steps (masked lanes read record 0); walk_consume turns them into LDS atomics a tile later.
    // Round 4 (profiles/r04_gram_knockouts.log: write-out 23.5 ms + entry loads 22.3 ms + LDS atomics 23.4 ms = the 69.3 ms of
    // the kernel -- three phases that did not overlap at all):
    //  * products and LDS addresses are computed in straight-line code and PINNED in registers before the first conditional
    //    atomic.  With the multiply inside the `if`, every block started with s_waitcnt lgkmcnt(0) (the lane-shuffled row
    //    scalar is older than an unknown number of atomics of earlier blocks): sixteen atomics per wave, each waiting for
    //    the one before; the compiler had also sunk the value half of the first entry's load into its block (a dependent
    //    round trip per tile);
    //  * the entries of the NEXT tile are requested before this tile's write-out (see the tile loop).
    struct Walk {
        int32_t ln[NSTEP], sk[NSTEP];
        T av[NSTEP];
        SpEntry<T> e[HH][NSTEP];
    };
    auto walk_issue = [&](const LevelC& c, Walk& w) {
        const int32_t hs = (int32_t)c.o0, hlen = c.valid ? (int32_t)(c.o1 - c.o0) : 0;  // 0 for lanes past the end / positions that hold nothing
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {
            const int src = k * RPS + grp;
            w.sk[k] = __shfl(hs, src);
            w.ln[k] = __shfl(hlen, src);
            w.av[k] = __shfl(c.a, src);
        }
#pragma unroll
        for (int k = 0; k < NSTEP; ++k)
#pragma unroll
            for (int hh = 0; hh < HH; ++hh) w.e[hh][k] = rec[sub + hh * SUB < w.ln[k] ? w.sk[k] + sub + hh * SUB : 0];
    };
    auto walk_consume = [&](const Walk& w, int64_t j_lo, int64_t tile_lo) {
        const int32_t jl = (int32_t)j_lo, tl = (int32_t)tile_lo;  // column indices are 32-bit
        T prod[HH][NSTEP];
        int32_t at[HH][NSTEP];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                const bool ok = sub + hh * SUB < w.ln[k] && w.e[hh][k].c >= jl;
                at[hh][k] = ok ? w.e[hh][k].c - tl : -1;
                prod[hh][k] = vt<T>::mul(w.av[k], w.e[hh][k].v);
                pin_vgpr(at[hh][k]);
                pin_vgpr(prod[hh][k]);
            }
        // ds_add_f32 runs at ~3 cycles per active lane on gfx950 (common.hpp, lds_accum): the first product of a cell goes in
        // with an integer compare-and-swap against zero -- all of a batch in flight together --, a product that found its cell
        // taken (~11 % at this fill) with a second swap on the value it saw, and only what loses that race too with the
        // floating-point atomic.  (fp64: the swaps are no-ops that report "taken"; ds_add_f64 is fast.)
        unsigned was[HH][NSTEP];
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                was[hh][k] = 0u;
                if (at[hh][k] >= 0) was[hh][k] = lds_accum_swap(&acc[at[hh][k]], prod[hh][k]);
            }
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)  // the returned patterns are looked at only HERE, after every swap has been issued (left to itself
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) pin_vgpr(was[hh][k]);  // the compiler compares inside each block: a wait per swap)
        unsigned got[HH][NSTEP];  // what the second swap saw (== was: the product is in)
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) {
                got[hh][k] = was[hh][k];
                if (was[hh][k] != 0u) got[hh][k] = lds_accum_retry(&acc[at[hh][k]], prod[hh][k], was[hh][k]);
            }
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int k = 0; k < NSTEP; ++k) pin_vgpr(got[hh][k]);
#pragma unroll
        for (int hh = 0; hh < HH; ++hh)
#pragma unroll
            for (int k = 0; k < NSTEP; ++k)
                if (got[hh][k] != was[hh][k]) atomic_accum(&acc[at[hh][k]], prod[hh][k]);
#pragma unroll
        for (int k = 0; k < NSTEP; ++k) {  // slices longer than HH * SUB entries
            for (int q = sub + HH * SUB; q < w.ln[k]; q += SUB) {
                const SpEntry<T> x = rec[w.sk[k] + q];
                if (x.c >= jl) lds_accum(&acc[x.c - tl], vt<T>::mul(w.av[k], x.v));
            }
        }
    };

    // ---- prologue: fill the pipeline (blocking chains, once per workgroup) ----
    // At the top of the iteration for tile k:  w = entries of tile k (in flight),  cv = level C of tile k + 1,
    // bv = level B of tile k + 2,  a3 = level A of tile k + 3  -- all requested during the iteration for tile k - 1.
    Pos p0, p1, p2, p3, p4;
    if (queue) {
        if (tid == 0)
            for (int k = 0; k < 5; ++k) s_pull[k] = atomicAdd(&queue[xq], 1ull);
        __syncthreads();
        p0 = pos_of(s_pull[0]);
        p1 = pos_of(s_pull[1]);
        p2 = pos_of(s_pull[2]);
        p3 = pos_of(s_pull[3]);
        p4 = pos_of(s_pull[4]);
        __syncthreads();
    } else {
        p0 = seek(blockIdx.x);
        p1 = after(p0);
        p2 = after(p1);
        p3 = after(p2);
        p4 = after(p3);
    }
    LevelA a0 = issue_a(p0), a1 = issue_a(p1), a2 = issue_a(p2), a3 = issue_a(p3);
    LevelC cv;
    LevelB bv;
    Walk w;
    {
        const LevelB b0 = issue_b(a0, p0.g), b1 = issue_b(a1, p1.g);
        bv = issue_b(a2, p2.g);
        const LevelC c0 = issue_c(b0, p0.g);
        cv = issue_c(b1, p1.g);
        walk_issue(c0, w);
    }
    for (int k = tid; k < TILE; k += nthreads) acc[k] = vt<T>::zero();
    __syncthreads();
    while (p0.ok) {
        const int64_t i = p0.i, g = p0.g;
        const int64_t tile_lo = g * TILE;
        const int64_t j_lo = i > tile_lo ? i : tile_lo;
        const int64_t j_hi = tile_lo + TILE < n ? tile_lo + TILE : n;
        // (1) this tile's products into LDS: the entries were requested a tile ago, before the previous tile's write-out
        walk_consume(w, j_lo, tile_lo);
        for (int64_t base = a0.t0 + (int64_t)(wave + nwaves) * WAVE; base < a0.t1; base += (int64_t)nwaves * WAVE) {
            // further rows of a long list (more than 64 * nwaves nonzeros in column i): fetched here, the chain exposed
            LevelA ax;
            ax.t0 = base - (int64_t)wave * WAVE;  // so that issue_b addresses `base`
            ax.t1 = a0.t1;
            const LevelB bx = issue_b(ax, g);
            const LevelC cx = issue_c(bx, g);
            Walk x;
            walk_issue(cx, x);
            walk_consume(x, j_lo, tile_lo);
        }
        // (2) one level of each of the four tiles ahead, every load issued here -- BEFORE the barrier and the stores of the
        //     write-out, so that their latency runs under the drain of those stores -- and consumed in the next iteration
        walk_issue(cv, w);         // E of tile k + 1
        cv = issue_c(bv, p2.g);    // C of tile k + 2
        bv = issue_b(a3, p3.g);    // B of tile k + 3
        const LevelA a4 = issue_a(p4);
        unsigned long long pulled = 0;  // the pair five tiles ahead: requested here, looked at after the write-out
        if (queue && tid == 0) pulled = atomicAdd(&queue[xq], 1ull);
        __syncthreads();
        // (3) the finished tile out, and back to zero
        syrkd_flush_tile(acc, C + (i - row0) * c_rs, c_cs, j_lo, j_hi, tile_lo, beta, beta_zero, tid, nthreads);
        if (queue && tid == 0) s_pull[0] = pulled;
        __syncthreads();
        p0 = p1;
        p1 = p2;
        p2 = p3;
        p3 = p4;
        p4 = queue ? pos_of(s_pull[0]) : after(p4);
        a0 = a1;
        a1 = a2;
        a2 = a3;
        a3 = a4;
    }
}

template <typename T>
static int syrkd_generic(int op, mi_sparse_matrix_t A, T alpha, T beta, T* C, int layout, int64_t ldc, int64_t row0 = 0,
                         int64_t row1 = -1)
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
        if (row1 < 0) row1 = n;
        if (row0 < 0 || row0 > row1 || row1 > n) fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad output row range [%lld, %lld) of %lld", (long long)row0, (long long)row1, (long long)n);
        const int64_t nr = row1 - row0;  // rows of the output block C points at
        if (n == 0 || nr == 0) return;
        if (!C) fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL output array");
        if (ldc < (layout == MI_SPARSE_LAYOUT_ROW_MAJOR ? n : nr)) fail(MI_SPARSE_STATUS_INVALID_VALUE, "ldc too small");
        Context& c = ctx();
        c.scratch_reset();
        // X = A (A^T A) or A^T (A A^T);  t = CSR of X^T, x = CSR of X
        Csr& x = aat ? need_csrT(h) : need_csr(h);
        Csr& t = aat ? need_csr(h) : need_csrT(h);
        const bool row_major = layout == MI_SPARSE_LAYOUT_ROW_MAJOR;
        const int64_t c_rs = row_major ? ldc : 1, c_cs = row_major ? 1 : ldc;
        Staged sc;
        const int beta_zero = vt<T>::is_zero(beta) ? 1 : 0;
        // lower triangle must survive the round trip
        sc.stage_in(C, sizeof(T) * (size_t)(row_major ? (nr - 1) * ldc + n : (n - 1) * ldc + nr), true);
        T* dC = static_cast<T*>(sc.dev);
        // wide outputs (more than one 64 KiB tile per row): 128 KiB tiles, one workgroup per CU
        const bool wide = n > (int64_t)syrkd_tile<T, 64>() && options().gram_tile_kb != 64;
        // 152 KiB tiles (all a workgroup can have next to nothing else) when they save a tile per output row: every (selected
        // row, tile) pair is a line request, so the literal configs[3] (262 144 columns) runs 7 tiles per row instead of 8
        bool xwide = wide && options().gram_tile_kb != 128 && options().gram_sliced != 0 &&
                     (options().gram_tile_kb == 152 || ceil_div(n, (int64_t)syrkd_tile<T, 152>()) < ceil_div(n, (int64_t)syrkd_tile<T, 128>()));
        int64_t tile = 0, tiles_per_row = 0, nblocks = 0;
        auto set_tile = [&]() {
            tile = xwide ? syrkd_tile<T, 152>() : wide ? syrkd_tile<T, 128>() : syrkd_tile<T, 64>();
            tiles_per_row = ceil_div(n, tile);
            nblocks = ceil_div(nr, 8) * 8 * tiles_per_row;  // 8 rows (one per XCD) x all their tiles per group
        };
        set_tile();
        if (nblocks > 2000000000) fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "gram output too large for one launch");
        // sliced walk when the rows of X are sorted (always, for a transpose built here) and a row's share of one tile is
        // short: with long slices the whole-row walk (64 lanes per row) is the faster one (2^20 x 65 536, 64 per row:
        // 15.8 vs 19.3 ms; literal configs[3], 8 per slice: 141 vs 116 ms)
        bool sliced = options().gram_sliced != 0 && x.nnz > 0;
        if (sliced && options().gram_sliced == 1 && x.nnz / (x.rows > 0 ? x.rows : 1) / tiles_per_row > 12) sliced = false;
        if (sliced && !cache_get(x.sorted)) {
            if (!rows_sorted(x)) sliced = false;  // (a positive answer is recorded by rows_sorted)
        }
        if (x.nnz >= ((int64_t)1 << 31) - 64) sliced = false;  // 32-bit absolute positions in the slice table
        size_t need = sizeof(int32_t) * (size_t)x.rows * (size_t)(tiles_per_row + 1);
        if (need > ((size_t)16 << 30)) sliced = false;  // slice table out of proportion (very tall X, very wide output)
        if (xwide && !sliced) {  // the whole-row walk keeps its 128 KiB tiles
            xwide = false;
            set_tile();
            need = sizeof(int32_t) * (size_t)x.rows * (size_t)(tiles_per_row + 1);
        }
        // persistent grid: gram_persistent workgroups per LDS slot (one 128 KiB or two 64 KiB tiles per CU), a multiple of 8
        int64_t grid = nblocks;
        const int64_t persistent = options().gram_persistent >= 0 ? options().gram_persistent : (sliced ? 1 : 4);
        if (persistent > 0) {
            c.ensure();
            const int64_t slots = (int64_t)c.cus * (wide ? 1 : 2) * persistent;
            // the list position of a workgroup advances by grid / 8 (row, tile) pairs per step: coprime with the tiles
            // per row, or a workgroup would see the same tile index for ever (tile 0 always full, the last always empty)
            int64_t g = slots / 8 > 0 ? slots / 8 : 1;
            auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t r = a % b; a = b; b = r; } return a; };
            while (g > 1 && gcd(g, tiles_per_row) != 1) --g;
            if (g * 8 < grid) grid = g * 8;
        }
        // the sliced walk's two cached tables; if the device has no room for them (a 256 GiB output leaves little), the
        // whole-row walk runs instead
        const int32_t* off = nullptr;
        const SpEntry<T>* rec = nullptr;
        if (sliced) {
            try {
                if (tiles_per_row > 1 && (x.gram_off_w != tile || !x.gram_off.p)) {
                    x.gram_off_w = 0;
                    x.gram_off.alloc(need);
                    MI_LAUNCH(k_gram_offsets, dim3((unsigned)ceil_div(x.rows * (tiles_per_row + 1), 256)), dim3(256), c.stream,
                              x.rows, tiles_per_row, tile, (const int64_t*)x.ptr, (const int32_t*)x.col, x.gram_off.as<int32_t>());
                    x.gram_off_w = tile;
                }
                if (!x.gram_rec.p) {  // packed (column, value) records of X
                    x.gram_rec.alloc(sizeof(SpEntry<T>) * (size_t)x.nnz);
                    const int64_t pg = ceil_div(x.nnz, 256) < 65536 ? ceil_div(x.nnz, 256) : 65536;
                    MI_LAUNCH((k_gram_pack<T>), dim3((unsigned)pg), dim3(256), c.stream, x.nnz, (const int32_t*)x.col,
                              (const T*)x.val, x.gram_rec.as<SpEntry<T>>());
                }
                if (tiles_per_row > 1) off = x.gram_off.as<int32_t>();
                rec = x.gram_rec.as<SpEntry<T>>();
            } catch (const status_error&) {
                clear_error();
                x.gram_rec.release();
                sliced = false;
                off = nullptr;
                if (xwide) {
                    xwide = false;
                    set_tile();
                }
                if (persistent > 0 && options().gram_persistent < 0) {  // the whole-row walk's default grid
                    const int64_t slots = (int64_t)c.cus * (wide ? 1 : 2) * 4;
                    int64_t g = slots / 8 > 0 ? slots / 8 : 1;
                    auto gcd = [](int64_t a, int64_t b) { while (b) { const int64_t r = a % b; a = b; b = r; } return a; };
                    while (g > 1 && gcd(g, tiles_per_row) != 1) --g;
                    grid = g * 8 < nblocks ? g * 8 : nblocks;
                }
            }
        }
#define MI_SYRKD_ARGS                                                                                               \
    (const int64_t*)t.ptr, (const int32_t*)t.col, (const T*)t.val, (const int64_t*)x.ptr, (const int32_t*)x.col,   \
        (const T*)x.val
        // slice bounds travelling with the entries of X^T (rows of X of at most 255 entries, at most 11 tiles per row)
        const GramHead* head = nullptr;
        if (sliced && off && tiles_per_row <= GRAM_HEAD_MAXG && options().gram_heads) {
            if (cache_get(x.gram_max_row) < 0) {
                long long* dm = static_cast<long long*>(c.scratch_alloc(sizeof(long long)));
                MI_HIP_CHECK(hipMemsetAsync(dm, 0, sizeof(long long), c.stream));
                MI_LAUNCH(k_gram_max_row, dim3((unsigned)(ceil_div(x.rows, 256) < 4096 ? ceil_div(x.rows, 256) : 4096)), dim3(256),
                          c.stream, x.rows, (const int64_t*)x.ptr, dm);
                long long hm = 0;
                MI_HIP_CHECK(hipMemcpyAsync(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost, c.stream));
                MI_HIP_CHECK(hipStreamSynchronize(c.stream));
                cache_set(x.gram_max_row, (int64_t)hm);
            }
            if (cache_get(x.gram_max_row) <= 255) {
                bool have = t.gram_head_w == tile && t.gram_head.p;
                if (!have) {
                    try {  // 16 bytes per nonzero: with a 256 GiB output on the device there may be no room -- the table still works
                        t.gram_head.alloc(sizeof(GramHead) * (size_t)t.nnz);
                        have = true;
                    } catch (const status_error&) {
                        clear_error();
                        t.gram_head_w = 0;
                    }
                    if (have) {
                        const int64_t hg = ceil_div(t.nnz, 256) < 65536 ? ceil_div(t.nnz, 256) : 65536;
                        MI_LAUNCH(k_gram_heads, dim3((unsigned)hg), dim3(256), c.stream, t.nnz, tiles_per_row,
                                  (const int32_t*)t.col, off, t.gram_head.as<GramHead>());
                        t.gram_head_w = tile;
                    }
                }
                if (have) head = t.gram_head.as<GramHead>();
            }
        }
        const int mode = !sliced ? -1 : head ? 2 : off ? 1 : 0;
        // option "deterministic": ONE wave per workgroup.  A tile is owned by one workgroup; with a single wave its products
        // reach the LDS in program order (and the lanes of one instruction in the hardware's fixed lane order), so the
        // floating-point sums are formed in the same order on every run -- at the price of 1/16 of the waves.
        const bool det = options().deterministic != 0;
        note_kernel("mi::%s<%s, TKB=%d%s%.0d>", sliced ? "k_syrkd_sliced" : "k_syrkd_lds", type_name<T>(), xwide ? 152 : wide ? 128 : 64,
                    sliced ? ", MODE=" : "", sliced ? mode + 0 : 0);
        if (sliced) {
            // one counter per XCD (workgroup b pulls from counter b % 8); persistent grids only
            unsigned long long* queue = nullptr;
            if (options().gram_queue && grid < nblocks && grid % 8 == 0) {
                queue = static_cast<unsigned long long*>(c.scratch_alloc(sizeof(unsigned long long) * 8));
                MI_HIP_CHECK(hipMemsetAsync(queue, 0, sizeof(unsigned long long) * 8, c.stream));
            }
#define MI_SLICED(TKB_, MODE_, THREADS_)                                                                               \
    MI_LAUNCH((k_syrkd_sliced<T, TKB_, MODE_>), dim3((unsigned)grid), dim3(det ? WAVE : THREADS_), c.stream, n, row0, row1, \
              tiles_per_row, (const int64_t*)t.ptr, (const int32_t*)t.col, (const T*)t.val, (const int64_t*)x.ptr, rec, off, head, \
              dC, c_rs, c_cs, alpha, beta, beta_zero, nblocks, queue)
            if (xwide) {
                if (mode == 2) MI_SLICED(152, 2, 1024); else if (mode == 1) MI_SLICED(152, 1, 1024); else MI_SLICED(152, 0, 1024);
            } else if (wide) {
                if (mode == 2) MI_SLICED(128, 2, 1024); else if (mode == 1) MI_SLICED(128, 1, 1024); else MI_SLICED(128, 0, 1024);
            } else {
                if (mode == 2) MI_SLICED(64, 2, 512); else if (mode == 1) MI_SLICED(64, 1, 512); else MI_SLICED(64, 0, 512);
            }
#undef MI_SLICED
        } else if (wide) {
            MI_LAUNCH((k_syrkd_lds<T, 128>), dim3((unsigned)grid), dim3(det ? WAVE : 1024), c.stream, n, row0, row1, tiles_per_row,
                      MI_SYRKD_ARGS, dC, c_rs, c_cs, alpha, beta, beta_zero, nblocks);
        } else {
            MI_LAUNCH((k_syrkd_lds<T, 64>), dim3((unsigned)grid), dim3(det ? WAVE : 512), c.stream, n, row0, row1, tiles_per_row,
                      MI_SYRKD_ARGS, dC, c_rs, c_cs, alpha, beta, beta_zero, nblocks);
        }
#undef MI_SYRKD_ARGS
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
mi_sparse_status_t mi_sparse_s_syrkd_rows(int op, mi_sparse_matrix_t A, float alpha, float beta, float* C, int layout,
                                          int64_t ldc, int64_t row0, int64_t row1)
{
    return mi::syrkd_generic<float>(op, A, alpha, beta, C, layout, ldc, row0, row1);
}
mi_sparse_status_t mi_sparse_d_syrkd_rows(int op, mi_sparse_matrix_t A, double alpha, double beta, double* C, int layout,
                                          int64_t ldc, int64_t row0, int64_t row1)
{
    return mi::syrkd_generic<double>(op, A, alpha, beta, C, layout, ldc, row0, row1);
}

}  // extern "C"
