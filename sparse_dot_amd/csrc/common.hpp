// common.hpp -- internals shared by every translation unit of libmi_sparse.so.
//
// Host side: status / error plumbing, per-thread context (device, stream, scratch arena),
// device-buffer RAII, pointer-location detection, the sparse handle.
// Device side: value-type traits (real + complex arithmetic), 16-byte vector wrapper, wave
// helpers.  gfx950 only: wavefront = 64 lanes everywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mi_sparse.h"

namespace mi {
// Launch helper: converts the arguments to the kernel's parameter types (so call sites need no
// casts).
[[noreturn]] void launch_too_large(unsigned long long threads);  // throws (runtime.hip)

template <typename... KArgs, typename... Args>
inline void launch_k(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args&&... args)
{
    // HIP runs at most 2^32 - 1 threads per grid dimension and silently drops the rest
    if ((unsigned long long)grid.x * block.x >= (1ull << 32)) launch_too_large((unsigned long long)grid.x * block.x);
    if (grid.y > 65535u || grid.z > 65535u) launch_too_large((unsigned long long)(grid.y > grid.z ? grid.y : grid.z));
    kernel<<<grid, block, smem, stream>>>(static_cast<KArgs>(args)...);
}
}  // namespace mi
#define MI_LAUNCH(kernel, grid, block, stream, ...) ::mi::launch_k(kernel, grid, block, 0, stream, __VA_ARGS__)
#define MI_LAUNCH_SMEM(kernel, grid, block, smem, stream, ...) \
    ::mi::launch_k(kernel, grid, block, smem, stream, __VA_ARGS__)
#define MI_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]

namespace mi {

constexpr int WAVE = 64;

// register-vector types of the buffer-load / MFMA builtins
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// complex value type (layout-compatible with mi_complex8 / mi_complex16 and numpy complex)
// ------------------------------------------------------------------------------------------------
template <typename R>
struct cx {
    R re, im;
};
using cfloat = cx<float>;
using cdouble = cx<double>;

template <typename T>
struct vt {  // real types
    using real = T;
    static constexpr bool is_complex = false;
    static __host__ __device__ __forceinline__ T zero() { return T(0); }
    static __host__ __device__ __forceinline__ T one() { return T(1); }
    static __host__ __device__ __forceinline__ T mul(T a, T b) { return a * b; }
    static __host__ __device__ __forceinline__ T add(T a, T b) { return a + b; }
    // explicit fused multiply-add: one rounding, independent of how a given instantiation is
    // scheduled, so every kernel variant produces the same bits
    static __host__ __device__ __forceinline__ T fma(T a, T b, T c) { return fma_(a, b, c); }
    static __host__ __device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static __host__ __device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static __host__ __device__ __forceinline__ T conj(T a) { return a; }
    static __host__ __device__ __forceinline__ bool is_zero(T a) { return a == T(0); }
};
template <typename R>
struct vt<cx<R>> {
    using real = R;
    using T = cx<R>;
    static constexpr bool is_complex = true;
    static __host__ __device__ __forceinline__ T zero() { return T{R(0), R(0)}; }
    static __host__ __device__ __forceinline__ T one() { return T{R(1), R(0)}; }
    static __host__ __device__ __forceinline__ T mul(T a, T b)
    {
        return T{vt<R>::fma(a.re, b.re, -(a.im * b.im)), vt<R>::fma(a.re, b.im, a.im * b.re)};
    }
    static __host__ __device__ __forceinline__ T add(T a, T b) { return T{a.re + b.re, a.im + b.im}; }
    static __host__ __device__ __forceinline__ T fma(T a, T b, T c)
    {
        return T{vt<R>::fma(-a.im, b.im, vt<R>::fma(a.re, b.re, c.re)),
                 vt<R>::fma(a.im, b.re, vt<R>::fma(a.re, b.im, c.im))};
    }
    static __host__ __device__ __forceinline__ T conj(T a) { return T{a.re, -a.im}; }
    static __host__ __device__ __forceinline__ bool is_zero(T a) { return a.re == R(0) && a.im == R(0); }
};

template <typename T>
struct type_char;
template <>
struct type_char<float> {
    static constexpr char value = 's';
};
template <>
struct type_char<double> {
    static constexpr char value = 'd';
};
template <>
struct type_char<cfloat> {
    static constexpr char value = 'c';
};
template <>
struct type_char<cdouble> {
    static constexpr char value = 'z';
};

inline size_t value_bytes(char t) { return t == 's' ? 4 : (t == 'd' || t == 'c') ? 8 : 16; }

// V consecutive values moved with one memory instruction (16 B when V * sizeof(T) == 16)
template <typename T, int V>
struct alignas(V * sizeof(T) >= 16 ? 16 : V * sizeof(T)) vec {
    T v[V];
};

// one sparse entry as a record -- column + value side by side, so that ONE 8-byte (float) / 16-byte (double, complex
// float) load or LDS read fetches both.  Staged nonzeros of A in the SpMM kernel; packed copy of X for the dense gram.
template <typename T>
struct alignas(sizeof(T) >= 16 ? 16 : 8) SpEntry {
    int32_t c;
    T v;
};

// streaming (touched-once) loads: non-temporal for the types the builtin takes, plain for the complex structs
template <typename T>
__device__ __forceinline__ T nt_load(const T* p)
{
    if constexpr (std::is_arithmetic<T>::value) return __builtin_nontemporal_load(p);
    else return *p;
}

// streaming stores (written once, not read again by the kernel): one element, or 16 bytes from `src` to a 16-byte
// aligned `dst`
template <typename T>
__device__ __forceinline__ void nt_store(T* dst, T v)
{
    __builtin_nontemporal_store(v, dst);
}
template <typename T>
__device__ __forceinline__ void nt_store16(T* dst, const T* src)
{
    u32x4 v;
    __builtin_memcpy(&v, src, 16);  // registers -> one 16-byte vector (no memory traffic after optimisation)
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
}

// value of lane `src` when `src` is the SAME in every lane: v_readlane_b32 -- a scalar result, nothing goes through the
// LDS queue (as __shfl = ds_bpermute it is an LDS-pipeline instruction per 32 bits and the value stays in a VGPR)
template <typename T>
__device__ __forceinline__ T lane_bcast(T v, int src)
{
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "lane_bcast: 4- or 8-byte values");
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) w[k] = __builtin_amdgcn_readlane(w[k], src);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// Position of this lane among the lanes of the wave whose `flag` is set, and their number (all 64 lanes call it together).
__device__ __forceinline__ int wave_rank(bool flag, int& total)
{
    const unsigned long long m = __ballot(flag);
    total = __popcll(m);
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// A 64-bit word other WORKGROUPS publish and poll (decoupled look-back of the one-pass SpGEMM): relaxed, agent scope -- the
// word carries its own status bits, nothing else is ordered by it.
__device__ __forceinline__ unsigned long long agent_load(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(unsigned long long* p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the lanes run in lockstep, an LDS fence is all
// the hardware needs
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// "this value is needed HERE": an empty asm the optimiser must feed with the value in a VGPR.  Stops the sinking of loads /
// multiplies into the conditional block that uses them -- where the wait-count pass, not knowing how many LDS / memory
// operations earlier conditional blocks have issued, falls back to s_waitcnt 0 in every block (gram.hip, round 4).
template <typename T>
__device__ __forceinline__ void pin_vgpr(T& x)
{
    asm volatile("" : "+v"(x));
}

// atomic accumulate (LDS or global).  Real types map to the hardware float / double atomic add
// (-munsafe-fp-atomics); complex does the two components independently.
template <typename T>
__device__ __forceinline__ void atomic_accum(T* p, T x)
{
    atomicAdd(p, x);
}
template <typename R>
__device__ __forceinline__ void atomic_accum(cx<R>* p, cx<R> x)
{
    atomicAdd(&p->re, x.re);
    atomicAdd(&p->im, x.im);
}

// Accumulate into an LDS cell that STARTED AT ZERO (gram tiles, SpGEMM hash values).
// gfx950 executes ds_add_f32 at ~3 cycles PER ACTIVE LANE -- 193 cycles for a full wave instruction -- while ds_add_f64 takes
// 21, the integer ds_add_u32 9 and a compare-and-swap with return ~13 (tools/probes/lds_atomic_probe.hip, profiles/
// r04_lds_atomic_probe.log); the dense fp32 gram kernel spent a third of its time in that instruction.  For fp32 the
// floating-point atomic is therefore the LAST resort:
//   1. compare-and-swap of the zero bit pattern against the product's bits (the first product of a cell needs no addition:
//      0 + x = x exactly);
//   2. the cell was taken: compare-and-swap of the value just seen against (seen + x);
//   3. lost that race too (several lanes on one cell): ds_add_f32, which serialises in hardware.
// Every interleaving gives what a sequence of atomic additions gives.  fp64 keeps its (fast) atomic.
template <typename T>
struct lds_word {  // bit pattern of an fp32 cell; a dummy for every other type
    using type = unsigned;
};
// step 1 for a batch: returns the pattern the cell held -- 0: the product is in.  Nothing has to look at the result before
// the other swaps of the batch have been issued.  Types without the swap report "taken".
template <typename T>
__device__ __forceinline__ unsigned lds_accum_swap(T* p, T x)
{
    (void)p;
    (void)x;
    return 1u;
}
// step 2: returns the pattern the cell held at the second swap -- equal to `seen`: the product is in
template <typename T>
__device__ __forceinline__ unsigned lds_accum_retry(T* p, T x, unsigned seen)
{
    (void)p;
    (void)x;
    return ~seen;
}
template <>
__device__ __forceinline__ unsigned lds_accum_swap<float>(float* p, float x)
{
    return atomicCAS(reinterpret_cast<unsigned*>(p), 0u, __float_as_uint(x));
}
template <>
__device__ __forceinline__ unsigned lds_accum_retry<float>(float* p, float x, unsigned seen)
{
    return atomicCAS(reinterpret_cast<unsigned*>(p), seen, __float_as_uint(__uint_as_float(seen) + x));
}
// one product at a time
template <typename T>
__device__ __forceinline__ void lds_accum(T* p, T x)
{
    const unsigned seen = lds_accum_swap(p, x);
    if (seen != 0u && lds_accum_retry(p, x, seen) != seen) atomic_accum(p, x);
}
template <typename R>
__device__ __forceinline__ void lds_accum(cx<R>* p, cx<R> x)
{
    lds_accum(&p->re, x.re);
    lds_accum(&p->im, x.im);
}

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();
void clear_error();

struct status_error {  // thrown inside the library, converted to a status code at the C boundary
    int status;
};

[[noreturn]] void fail(int status, const char* fmt, ...);

#define MI_HIP_CHECK(expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            (void)hipGetLastError();                                                                    \
            ::mi::fail(_e == hipErrorOutOfMemory ? MI_SPARSE_STATUS_ALLOC_FAILED                        \
                                                 : MI_SPARSE_STATUS_EXECUTION_FAILED,                   \
                       "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
        }                                                                                               \
    } while (0)

// run `body` and map exceptions to MKL-style status codes; never lets anything escape the C ABI
template <typename F>
inline int guarded(F&& body) noexcept
{
    try {
        clear_error();
        body();
        return MI_SPARSE_STATUS_SUCCESS;
    } catch (const status_error& e) {
        return e.status;
    } catch (const std::bad_alloc&) {
        set_error("host allocation failed");
        return MI_SPARSE_STATUS_ALLOC_FAILED;
    } catch (...) {
        set_error("unexpected internal exception");
        return MI_SPARSE_STATUS_INTERNAL_ERROR;
    }
}

// ------------------------------------------------------------------------------------------------
// per-thread context
// ------------------------------------------------------------------------------------------------
// Owning device allocation.  Blocks come from / return to a per-device cache (runtime.hip): HBM is
// expensive to map -- a fresh 100 GB result costs ~1 s in hipMalloc, and hipFree + hipMalloc of the
// same size several seconds -- so a released block is kept (up to half of the device memory,
// option "pool_max_mb") and handed to the next request of about the same size.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;  // usable size (>= the size asked for)
    int dev = -1;      // device the block lives on
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), dev(o.dev) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept
    {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            dev = o.dev;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n);  // throws ALLOC_FAILED; n == 0 still yields a valid (tiny) allocation
    void release();
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

void pool_trim();       // return every cached block to the driver
void pool_reset_cap();  // re-read pool_max_mb on the next release

struct Context {
    bool initialised = false;
    int device = -1;  // -1: adopt the calling thread's current HIP device at first use (mi_sparse_set_device overrides)
    int cus = 0;      // compute units of `device` (persistent-kernel grids)
    size_t total_bytes = 0;  // memory of `device` (asked once: hipMemGetInfo is a driver query)
    hipStream_t stream = nullptr;
    // grow-only scratch arena; kernels on one stream execute in order, so a later call may reuse
    // the arena as soon as it is enqueued behind the earlier one
    DevBuf scratch;
    size_t scratch_used = 0;
    std::vector<DevBuf> retired;  // old arenas kept alive until the next synchronise
    void ensure();                // lazy HIP init for this host thread
    void* scratch_alloc(size_t bytes);  // 256-B aligned slice, valid until scratch_reset()
    void scratch_reserve(size_t bytes);  // room for `bytes` more, grown once and exactly
    void scratch_release();              // synchronise and give the arena back
    void scratch_reset() { scratch_used = 0; }
    void sync();
};
Context& ctx();
size_t device_total_bytes();  // of the calling thread's device

// Large host <-> device copies of PAGEABLE caller memory (the reference's calling convention hands over numpy
// buffers): a plain hipMemcpy stages them through one pinned bounce buffer on one thread (~18 GB/s measured);
// here a few worker threads copy 4 MiB chunks to / from their own pinned slots and drive their own DMA
// streams, so the host-side memcpy runs in parallel and overlaps the PCIe transfer (runtime.hip, CopyEngine).
// copy_h2d: on return the source has been read and the calling thread's stream is ordered behind the transfers.
// copy_d2h: waits for the calling thread's stream, returns when the bytes are in `dst`.
void copy_h2d(void* dst_dev, const void* src_host, size_t n);
void copy_d2h(void* dst_host, const void* src_dev, size_t n);

// where does a caller pointer live?
enum class Loc { Host, Device };
Loc locate(const void* p);

// dense / index array as seen by kernels: either the caller's device pointer or a staged copy
struct Staged {
    void* dev = nullptr;   // pointer kernels use
    void* host = nullptr;  // caller's host pointer (nullptr when the caller passed device memory)
    size_t bytes = 0;
    DevBuf own;
    // in: copy host -> device now.  out: call copy_back() after the kernels.
    void stage_in(const void* p, size_t n, bool copy_contents);
    void copy_back();
};

// ------------------------------------------------------------------------------------------------
// the handle
// ------------------------------------------------------------------------------------------------
constexpr uint32_t HANDLE_MAGIC = 0x4d495350u;  // 'MISP'

// canonical device CSR: 64-bit row pointer, 32-bit column index, values of the handle's type
template <typename X>
inline X cache_get(const X& f) { return __atomic_load_n(&f, __ATOMIC_RELAXED); }
template <typename X>
inline void cache_set(X& f, X v) { __atomic_store_n(&f, v, __ATOMIC_RELAXED); }

struct Csr {
    int64_t rows = 0, cols = 0, nnz = 0;
    int64_t* ptr = nullptr;  // rows + 1
    int32_t* col = nullptr;  // nnz
    void* val = nullptr;     // nnz
    DevBuf ptr_own, col_own, val_own;  // storage when the library owns it (else aliases caller HBM)
    bool valid = false;
    // column indices known to be ascending inside every row.  rows_sorted() records a positive answer through a const
    // reference (the structure of a handle never changes -- for arrays aliased from the caller that is part of the contract,
    // include/mi_sparse.h): host threads sharing an operand may race on it, so these cached answers are only ever touched
    // through cache_get / cache_set (relaxed atomics; every writer stores the same value)
    mutable bool sorted = false;
    // generation of the ENTRY ORDER inside the rows: a fresh value (next_order_gen) whenever the entries are (re)laid out --
    // built, transposed into, re-sorted.  Anything that indexes per-entry tables by position (the staged product's B-row
    // extents, spgemm.hip) records it and checks it again before trusting those tables.
    // SpGEMM results (spgemm.hip): rows of more than range_min_len entries consist of consecutive runs of range_cap entries with
    // disjoint, ascending column sets (only the inside of a run is unordered) -- mi_sparse_order sorts run by run.  0: no such layout.
    int64_t range_cap = 0, range_min_len = 0;
    // SpGEMM results accumulated by rank (k_spgemm_rank): rows of more than sorted_min_len entries already have their columns
    // in increasing order -- mi_sparse_order leaves them alone.  0: nothing known.
    int64_t sorted_min_len = 0;
    uint64_t order_gen = 0;
    // dense gram (gram.hip): ABSOLUTE position of the first entry of every row at or right of each tile boundary, int32[rows * (cols / w + 2)], built on first
    // use for tile width gram_off_w (structure only: unaffected by set_values)
    DevBuf gram_off;
    int64_t gram_off_w = 0;
    // dense gram, sliced walk: packed (column, value) records of every entry in storage order (SpEntry<T>[nnz]); follows
    // the VALUES, so mi_sparse_?_set_values and mi_sparse_order drop it
    DevBuf gram_rec;
    // dense gram: as the CSR of X^T -- per entry (r, i) the start of row r in X's records and its entries left of every tile
    // boundary (GramHead, gram.hip), built for tile width gram_head_w; as the CSR of X -- its longest row (-1: not known yet)
    DevBuf gram_head;
    int64_t gram_head_w = 0;
    mutable int64_t gram_max_row = -1;
};

uint64_t next_order_gen();  // handle.hip: process-wide, never repeats

// block form kept next to the CSR expansion on handles created from BSR arrays: the SpMM block kernel (bsr.hip) reads it
struct Bsr {
    int64_t brows = 0, bcols = 0, bs = 0, nblocks = 0;
    int layout = MI_SPARSE_LAYOUT_ROW_MAJOR;  // storage order INSIDE a block
    int64_t* ptr = nullptr;                   // brows + 1
    int32_t* col = nullptr;                   // nblocks (block column)
    void* val = nullptr;                      // nblocks * bs * bs
    DevBuf ptr_own, col_own, val_own;
    bool valid = false;
};

struct HostExport {  // library-owned host copies handed out by export_* (valid until destroy)
    std::vector<char> ptr, col, val;
};

// Small page-locked host block + event: results of device-side analysis come back with an asynchronous copy
// and are looked at by a LATER call (hipEventQuery), so no executor ever waits for the inspector.
struct AsyncWord {
    int64_t* host = nullptr;  // 8 x int64, hipHostMalloc
    hipEvent_t ev = nullptr;
    bool pending = false;
    AsyncWord() = default;
    AsyncWord(const AsyncWord&) = delete;
    AsyncWord& operator=(const AsyncWord&) = delete;
    ~AsyncWord();
    void ensure();             // allocate on first use
    void post(const void* dev_src, size_t bytes, hipStream_t s);  // enqueue D2H + event
    bool ready();              // true once a posted copy has landed (non-blocking)
};

struct SpmmKpart;
struct SpmmPlan {  // nnz+row balanced partition for the SpMM kernel (see spmm.hip)
    int chunk = 0;
    int64_t nchunks = 0;
    DevBuf chunk_row;  // int32[nchunks + 1]
    DevBuf chunk_desc; // SpmmChunk[nchunks] (spmm.hip): first row / rows / nonzero range / carry flag of every chunk
    // static fix-up schedule: one task per long row that is cut across chunks -- (row, first chunk,
    // last chunk) whose carries are added, in chunk order, to the row its owner wrote.  Built on the
    // device in one pass; the count stays on the device (n_tasks_dev) and reaches the host
    // asynchronously -- until then the fix-up kernel is launched for the upper bound (nchunks).
    DevBuf tasks;        // int32[3 * nchunks]
    DevBuf n_tasks_dev;  // unsigned long long[2]: short tasks (front of the list), long ones (from its back)
    int64_t n_tasks = -1, n_tasks_long = -1;  // -1: not known on the host yet
    AsyncWord n_tasks_word;
    // hot / cold column tagging (see spmm.hip): copy of the column indices with bit 31 set on
    // entries whose column is NOT in the hot set that is meant to stay L2 resident.  The analysis
    // (sampled column histogram -> threshold -> tags) runs on the device, enqueued behind the SECOND
    // product of a handle (a single-use handle never pays for it) and is adopted by the first later
    // call that finds its decision word landed.
    int64_t uses = 0;              // products executed with this plan
    int64_t hot_rows_budget = -1;  // the budget (in B rows) the analysis ran / is running for; -1 = never
    int hot_state = 0;             // 0 none, 1 analysis enqueued, 2 decision known
    bool tagged = false;           // decision: use the tagged gather
    double hot_coverage = 0.0;     // fraction of nonzeros that fall on hot columns
    DevBuf col_tagged;             // int32[nnz]
    DevBuf hot_decision;           // int64[8] on the device: {flag, thr, nhot, covered, total}
    AsyncWord hot_word;
    // column-partitioned form of the LONG rows (SpmmKpart, spmm.hip): 0 not looked at yet, 1 declined (the matrix has no
    // long rows worth it), 2 ready.  Holds a copy of the VALUES: mi_sparse_?_set_values / mi_sparse_order drop it.
    int kpart_state = 0;
    std::shared_ptr<SpmmKpart> kpart;
    void reset_kpart()
    {
        kpart_state = 0;
        kpart.reset();
        uses = 0;
    }
    void reset_hot()
    {
        hot_rows_budget = -1;
        hot_state = 0;
        tagged = false;
        hot_coverage = 0.0;
        col_tagged.release();
    }
};

// Column-partitioned SpMM plan (round 5, spmm.hip).  Rows of at least `min_row` entries ("long" rows) are split by
// part(column) into P sub-rows; the sub-rows of partition p form the p-th block of rows of `cat` and are multiplied by the
// workgroups of ONE set of XCDs only, so that set's L2 only ever holds rows of B of partition p -- the eight 4 MB L2s keep
// P times more DISTINCT rows of B between them.  The price is one partial output row per (long row, partition), summed in a
// fixed order (k_kp_combine).  Every other row stays in `shrt` (all rows of A, the long ones empty) and is multiplied
// row-owned as before.
struct SpmmKpart {
    int P = 0;            // column partitions (8, 4 or 2); the dense operand is cut into 8 / P column slices on top
    int64_t min_row = 0;  // rows of at least this many entries are partitioned
    int64_t chunk = 0;    // spmm_chunk the chunk ranges cs[] were computed for
    int64_t n_long = 0, nnz_long = 0;
    Csr cat;              // rows = P * n_long (block p = partition p, rows in the order of rowid), cols = A's
    Csr shrt;             // rows = A's, entries of the short rows only
    DevBuf rowid;         // int32[n_long]: row of A behind every long row
    int64_t cs[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // chunks [cs[p], cs[p + 1]) of cat's plan belong to partition p
    SpmmPlan plan_cat, plan_short;
};

}  // namespace mi

struct mi_sparse_matrix {
    uint32_t magic = mi::HANDLE_MAGIC;
    char vtype = 's';       // 's' 'd' 'c' 'z'
    int index_bytes = 4;    // flavour the handle was created with (4 / 8)
    int64_t rows = 0, cols = 0;
    char origin = 'r';      // 'r' created as CSR, 'c' as CSC, 'b' as BSR, 'l' library result
    mi::Csr csr;            // CSR of A
    mi::Csr csrT;           // CSR of A^T  (== the CSC arrays of A)
    mi::Bsr bsr;            // block form (origin 'b' only)
    bool bsr_pristine = true;  // origin 'b': not ordered since creation -> export_bsr hands back the arrays as they were given
    // caller's HOST arrays, kept for mi_sparse_order's write-back (nullptr when device / result)
    void* user_col = nullptr;
    void* user_val = nullptr;
    int user_base = 0;
    mi::SpmmPlan plan, planT;
    mi::HostExport exp_csr, exp_csc, exp_bsr;
    int64_t result_bs = 0;  // block size a product of two BSR handles is exported with (mkl_sparse_?_export_bsr)
    std::shared_ptr<void> staged;  // result handles of the staged product (mi_sparse_sp2m): symbolic-phase state
    std::mutex mtx;  // guards lazy derivation of csr / csrT / plans
};

namespace mi {

mi_sparse_matrix* check_handle(mi_sparse_matrix_t h);  // throws NOT_INITIALIZED
Csr& need_csr(mi_sparse_matrix* h);                    // derive (transpose) if only csrT is there
Csr& need_csrT(mi_sparse_matrix* h);
bool rows_sorted(const Csr& a);  // column indices ascending inside every row (one pass over the indices unless known)
void transpose_csr(char vtype, const Csr& in, Csr& out, bool conj);  // out := in^T (stable, sorted rows)
void sort_csr(char vtype, Csr& a);                     // in-place order of every row
mi_sparse_matrix* new_result_handle(char vtype, int index_bytes, int64_t rows, int64_t cols);

// device-side exclusive scan of int64 counts (n entries) into out[n + 1]; returns the total
int64_t exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t n);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// dispatch a callable templated on the value type:  by_type(vtype, [&](auto tag){ using T = decltype(tag); ... })
template <typename F>
inline void by_type(char vtype, F&& f)
{
    switch (vtype) {
        case 's': f(float{}); break;
        case 'd': f(double{}); break;
        case 'c': f(cfloat{}); break;
        case 'z': f(cdouble{}); break;
        default: fail(MI_SPARSE_STATUS_INTERNAL_ERROR, "bad value type %d", (int)vtype);
    }
}

// options (mi_sparse_set_option)
struct Options {
    int64_t spmm_chunk = 256;      // work items (nnz + row ends) per wave in the SpMM kernel
    int64_t spmm_force_generic = 0;
    int64_t spmm_unroll = 4;       // 4 or 8 independent B-row loads in flight per lane
    int64_t spmm_hot_force = 0;    // tests: tag even tiny / unskewed matrices
    int64_t spmm_slices = 0;       // XCD-affine column slices of the dense operand: 0 = by row width (256-byte slices), else 1, 2, 4, 8
    int64_t spmm_hot_kb = 8192;    // bytes of hot B rows to keep L2 resident (0 disables hot/cold tagging)
    int64_t gemm_big_tiles = -1;   // dense x dense: products of at least this many 128 x 128 output tiles take the pipelined 128-tile kernel (-1: one per CU; 0: never)
    int64_t spmm_kpart = 1;        // column-partitioned long rows (SpmmKpart) from the third product of a handle on: 0 never, 1 when it pays, 2 always (tests)
    int64_t spmm_kpart_min_row = 64;   // ... rows of at least this many entries (the partial rows cost 2 x 8 row widths per long row: break-even ~53 entries; 48 / 64 / 96 / 128 / 160 on the headline matrix: 1.248 / 1.240 / 1.240 / 1.27 / 1.37 ms)
    int64_t spmm_kpart_chunk = 128;    // ... work items per wave of the two partitioned kernels (spmm_chunk for every other product): 128 / 256 = 1.217-1.229 / 1.234-1.240 ms
    int64_t spmm_kpart_parts = 8;  // ... column partitions: 8, 4 or 2 (x 1, 2, 4 column slices of the dense operand)
    int64_t spgemm_force_global = 0;
    int64_t spgemm_lds_parts = 1;    // big rows: LDS bitmap (symbolic) / hash-partitioned LDS classes (numeric)
    int64_t spgemm_part_log2s_bias = 0;  // tuning: +1 / -1 forces the larger / smaller table of the numeric big-row kernel
    int64_t spgemm_slice_table = 1;  // big rows, numeric: precompute the B-row slices of every (row, range) (k_part_slices)
    int64_t spgemm_slice_table_max = (int64_t)3 << 30;  // ... unless the table would exceed this many int32 entries
    int64_t spgemm_global_mode = 0;  // 0: one workgroup per row, L2-local atomics; 1: cooperative, agent-scope atomics
    int64_t spgemm_group = 1;        // short rows of B (<= 32 entries): one 16-lane group per selected row of B instead of the flat product list (a third of the instructions)
    int64_t spgemm_rank = 0;         // 1: big rows (sorted B, real or complex-float values): the symbolic phase keeps the row bitmaps, the numeric phase accumulates by rank (k_spgemm_rank) and the rows of C come out sorted; 0: range-partitioned LDS hash (k_spgemm_part) -- same kernel time on the literal configs[2] (round 4), without the 128 KiB per big row of stored bitmap
    int64_t sort_ranges = 1;         // mi_sparse_order on SpGEMM results: rows written range by range are sorted range by range (0: as any other long row)
    int64_t transpose_radix = 1;     // transposes of 2^21 entries and more: stable radix sort of the entries by column (no atomics, no per-row sort afterwards); 2: the same with tiles of 8192 entries; 0: histogram + atomic scatter + row sort
    int64_t transpose_radix_bits = 7;  // ... most bits of the column index per pass (4 .. 9): 2^18 columns = 3 passes of 6 bits (8.7 ms for 2.7e8 entries; 2 passes of 9 bits scatter 64-byte runs: 10.8 ms)
    int64_t transpose_lds_hist = 1;  // column histogram of a transpose through LDS ranges (>= 2^22 entries, <= 2^20 columns); 0: one global atomic per entry
    int64_t spgemm_narrow_ptr = 1;   // k_row_ub gathers B's row extents from an int32 copy of its row pointer made per call (nnz(B) < 2^31, >= 2^16 rows): half the table, twice the pointers per line
    int64_t spmmd_lds = 1;           // sparse x sparse -> dense: rows of the result built in LDS tiles and written once (0: zero fill + global atomics)
    int64_t spgemm_col_panels = 1;   // B wider than the LDS bitmap of the big-row path (~1.1 M columns) and rows of the product too long for the LDS hash: product by panels of 2^20 columns (0: global-memory hash)
    int64_t spgemm_sort_ingest = 1;  // B with unsorted rows and rows of the product too long for the LDS hash: multiply by a sorted copy of B (0: global-memory hash)
    int64_t spgemm_onepass = 1;      // products whose rows all fit the small LDS tables: ONE kernel (no symbolic pass), rows placed by a decoupled look-back; 0: always two phases
    int64_t spgemm_hub = 0;          // hub rows of a (full) product through dense LDS accumulators over popularity-ordered column blocks of a relabelled copy of B (spgemm_hub.inc): 0 never (default: measured at parity with the range path on the literal configs[2] -- 166 vs 154 ms, profiles/r06_spgemm_hub_steps.log), 1 when the product is large and skewed enough, 2 whenever a row is beyond the LDS hash classes (tests), 3 the relabelling alone (range path on the relabelled copy: 164 ms)
    int64_t spgemm_hub_min_products = (int64_t)1 << 28;  // ... option value 1: products (upper bound) from which the relabelled copies pay
    int64_t spgemm_hub_fill_pct = 20;  // ... a block is accumulated densely for the rows whose expected products fill at least this share of it
    int64_t spgemm_hub_acc_kb = 64;    // ... accumulators of one workgroup (widest block = this many KiB of values, at most 16384 columns)
    int64_t spgemm_hub_block_kb = 1 << 20;  // ... a block holds at most this many KiB of B's entries (narrower blocks where the columns are popular)
    int64_t pool_enable = 1;       // cache released device blocks for reuse (0: hipFree at once)
    int64_t pool_max_mb = -1;      // cap on cached bytes; -1 = half of the device memory
    int64_t trace_phases = 0;      // print host wall-clock per SpGEMM phase to stderr (diagnostics; synchronises)
    int64_t gram_heads = 1;        // sliced dense gram: slice bounds travel with the entries of X^T (rows of X <= 255 entries, <= 11 tiles per row); 0: per-row table
    int64_t gram_sliced = 1;       // dense gram: slice table + 8 lanes per selected row when rows of X are sorted and slices are short (<= 12 entries on average); 2: whenever sorted; 0: never
    int64_t gram_persistent = -1;   // dense gram: workgroups per LDS slot of the chip walking the tile list (0: one workgroup per tile; -1: 1 for the sliced walk, 4 for the whole-row walk)
    int64_t gram_tile_kb = 0;      // dense gram LDS tile: 0 = 152 KiB where that saves a tile per output row, else 128; 64 / 128 / 152 force
    int64_t bsr_native = 1;        // BSR handles x row-major dense: the block kernel (0: always the CSR expansion)
    int64_t staged_copies = 1;     // large pageable host <-> device copies through the parallel pinned stager (0: plain hipMemcpy)
    int64_t profile_events = 0;    // bracket the SpMM main kernel with hipEvents (diagnostics)
    int64_t spmm_plan_sync = 0;    // 1: run the hot / cold analysis synchronously inside the first product (tests, A/B tools)
    int64_t gram_queue = 1;        // sliced dense gram: (row, tile) pairs pulled in order from one counter per XCD (0: fixed stride per workgroup)
    int64_t deterministic = 0;     // 1: run-to-run bitwise reproducible SpGEMM / sparse + dense gram (SpMM / SpMV always are): fixed summation order, slower
};
struct Counters {
    double spmm_kernel_ms = 0.0;
    double spmm_kernel_launches = 0.0;
    double spmm_last_tagged = 0.0;    // 1 when the last SpMM used the hot / cold tagged gather
    double spmm_hot_coverage = 0.0;   // share of nonzeros on hot columns in the last SpMM's plan
    double spmm_last_slices = 1.0;    // column slices the last SpMM ran with
    double spmm_plan_ms = 0.0;        // host wall time spent building plans (partition + fix-up schedule), accumulated
    double spmm_plans_built = 0.0;
    double spmm_last_kpart = 0.0;     // column partitions the last SpMM ran its long rows with (0: row-owned only)
    double spmm_kpart_long_share = 0.0;  // share of the nonzeros in partitioned rows (last SpMM)
    double spgemm_hub_items = 0.0;    // (hub row, column block) pairs accumulated densely by SpGEMM products, accumulated
    double spgemm_panels = 0.0;       // column panels multiplied by SpGEMM products of a B wider than the LDS bitmap, accumulated
    double spmm_kpart_build_ms = 0.0; // host wall time spent building column-partitioned plans, accumulated
    double bsr_native_calls = 0.0;    // products served by the BSR block kernel
};
Counters& counters();  // per host thread
void note_kernel(const char* fmt, ...);
const char* last_kernel_name();  // records the dominant kernel of the current call (mi_sparse_get_last_kernel)
template <typename T>
inline const char* type_name()
{
    return type_char<T>::value == 's' ? "float" : type_char<T>::value == 'd' ? "double" : type_char<T>::value == 'c' ? "cfloat" : "cdouble";
}

// bsr.hip
template <typename T>
bool bsr_spmm_applicable(const Bsr& b, int layout, const T* B, int64_t N, int64_t ldb, const T* C, int64_t ldc);
template <typename T>
void bsr_spmm_device(const Bsr& b, T alpha, const T* B, int64_t N, int64_t ldb, T beta, T* C, int64_t ldc);
Options& options();

}  // namespace mi
