// hip_emu.hpp -- DEVELOPER-ONLY host emulation of the small HIP subset libmi_sparse's kernels use.
//
// Purpose: the build container has no GPU and a gpurun round trip takes minutes, so this header
// lets the *same kernel source* (csrc/*.hip compiled with g++ -DMI_HIP_EMU) execute on host
// threads to shake out indexing / synchronisation logic before spending GPU time.  It is NOT a
// CPU backend: it is never built by __graft_entry__.build(), never shipped, never loaded by the
// sparse_dot_amd package (which fails loudly without a HIP device), and it is orders of
// magnitude slower than anything usable.  One OS thread per GPU thread of a workgroup;
// workgroups run one after another; `__shared__` becomes a function-local static.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef void* hipStream_t;
struct hip_emu_event { double t; };
typedef hip_emu_event* hipEvent_t;
enum { hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum hipMemoryType { hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1, hipMemoryTypeManaged = 3, hipMemoryTypeUnregistered = 4 };
struct hipPointerAttribute_t { hipMemoryType type; };
struct hipDeviceProp_t { char name[64]; char gcnArchName[64]; int multiProcessorCount; size_t totalGlobalMem; };

namespace hip_emu {

struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int live = 0, count = 0;
    unsigned gen = 0;
    void reset(int n) { live = n; count = 0; }
    void arrive_and_wait()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned g = gen;
        if (++count >= live) { count = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return g != gen; });
    }
    void drop()
    {
        std::unique_lock<std::mutex> lk(m);
        --live;
        if (live > 0 && count >= live) { count = 0; ++gen; cv.notify_all(); }
    }
};

struct WaveState {
    Barrier bar;
    alignas(16) unsigned char slot[64][16];
};

struct State {
    dim3 grid, block;
    Barrier block_bar;            // __syncthreads
    Barrier done_bar;             // end of a workgroup (all threads, never dropped)
    std::vector<WaveState> waves;
    std::vector<char> dyn;
    std::mutex alloc_m;
    std::map<uintptr_t, size_t> allocs;
};
inline State& st() { static State s; return s; }

inline thread_local int t_tid = 0;

inline char* dyn_smem() { return st().dyn.data(); }

template <typename... KArgs>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, KArgs... args);

}  // namespace hip_emu

inline thread_local dim3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

namespace hip_emu {
template <typename... KArgs>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t smem, KArgs... args)
{
    State& s = st();
    const int nthreads = (int)(block.x * block.y * block.z);
    const long nblocks = (long)grid.x * grid.y * grid.z;
    if (nthreads <= 0 || nblocks <= 0) return;
    s.grid = grid; s.block = block;
    ::blockDim = block; ::gridDim = grid;
    s.dyn.assign(smem + 16, 0);
    const int nwaves = (nthreads + 63) / 64;
    s.waves = std::vector<WaveState>(nwaves);
    s.done_bar.reset(nthreads);
    auto body = [&](int tid) {
        t_tid = tid;
        ::threadIdx = dim3(tid % block.x, (tid / block.x) % block.y, tid / (block.x * block.y));
        for (long b = 0; b < nblocks; ++b) {
            if (tid == 0) {
                s.block_bar.reset(nthreads);
                for (int w = 0; w < nwaves; ++w) s.waves[w].bar.reset(std::min(64, nthreads - w * 64));
            }
            s.done_bar.arrive_and_wait();
            ::blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
            kernel(args...);
            s.block_bar.drop();
            s.waves[tid / 64].bar.drop();
            s.done_bar.arrive_and_wait();
        }
    };
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int t = 0; t < nthreads; ++t) th.emplace_back(body, t);
    for (auto& t : th) t.join();
}
}  // namespace hip_emu

inline void __syncthreads() { hip_emu::st().block_bar.arrive_and_wait(); }

template <typename T>
inline T __shfl_xor(T v, int mask)
{
    static_assert(sizeof(T) <= 16, "shuffle payload too large");
    hip_emu::WaveState& w = hip_emu::st().waves[hip_emu::t_tid / 64];
    const int lane = hip_emu::t_tid % 64;
    std::memcpy(w.slot[lane], &v, sizeof(T));
    w.bar.arrive_and_wait();
    T r = v;
    const int src = lane ^ mask;
    if (src >= 0 && src < 64) std::memcpy(&r, w.slot[src], sizeof(T));
    w.bar.arrive_and_wait();
    return r;
}

template <typename T>
inline T __shfl_up(T v, int delta)
{
    static_assert(sizeof(T) <= 16, "shuffle payload too large");
    hip_emu::WaveState& w = hip_emu::st().waves[hip_emu::t_tid / 64];
    const int lane = hip_emu::t_tid % 64;
    std::memcpy(w.slot[lane], &v, sizeof(T));
    w.bar.arrive_and_wait();
    T r = v;
    if (lane - delta >= 0) std::memcpy(&r, w.slot[lane - delta], sizeof(T));
    w.bar.arrive_and_wait();
    return r;
}

template <typename T>
inline T __shfl(T v, int src)
{
    static_assert(sizeof(T) <= 16, "shuffle payload too large");
    hip_emu::WaveState& w = hip_emu::st().waves[hip_emu::t_tid / 64];
    const int lane = hip_emu::t_tid % 64;
    std::memcpy(w.slot[lane], &v, sizeof(T));
    w.bar.arrive_and_wait();
    T r = v;
    if (src >= 0 && src < 64) std::memcpy(&r, w.slot[src], sizeof(T));
    w.bar.arrive_and_wait();
    return r;
}

// ---- atomics -------------------------------------------------------------------------------------
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
template <typename F>
inline F emu_atomic_fadd(F* p, F v)
{
    static std::mutex m;  // coarse but correct
    std::lock_guard<std::mutex> lk(m);
    F old = *p;
    *p = old + v;
    return old;
}
inline float atomicAdd(float* p, float v) { return emu_atomic_fadd(p, v); }
inline double atomicAdd(double* p, double v) { return emu_atomic_fadd(p, v); }
inline int atomicCAS(int* p, int cmp, int val)
{
    __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int atomicMin(int* p, int v)
{
    int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
inline long long atomicMax(long long* p, long long v)
{
    long long old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

// ---- runtime API ---------------------------------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 2; return hipSuccess; }  // tiny persistent grids
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = size_t(8) << 30; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
    std::snprintf(p->name, sizeof(p->name), "host-thread emulation");
    std::snprintf(p->gcnArchName, sizeof(p->gcnArchName), "none");
    p->multiProcessorCount = 0;
    p->totalGlobalMem = 0;
    return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n)
{
    *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::memset(*p, 0xCD, n);  // poison: catches reads of uninitialised device memory
    std::lock_guard<std::mutex> lk(hip_emu::st().alloc_m);
    hip_emu::st().allocs[(uintptr_t)*p] = n ? n : 1;
    return hipSuccess;
}
template <typename T>
inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p)
{
    {
        std::lock_guard<std::mutex> lk(hip_emu::st().alloc_m);
        hip_emu::st().allocs.erase((uintptr_t)p);
    }
    std::free(p);
    return hipSuccess;
}
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hip_emu_event{0.0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = 0.0; return hipSuccess; }  // launches are synchronous here
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    std::lock_guard<std::mutex> lk(hip_emu::st().alloc_m);
    auto& m = hip_emu::st().allocs;
    auto it = m.upper_bound((uintptr_t)p);
    if (it != m.begin()) {
        --it;
        if ((uintptr_t)p >= it->first && (uintptr_t)p < it->first + it->second) { a->type = hipMemoryTypeDevice; return hipSuccess; }
    }
    return hipErrorInvalidValue;
}

// wave-scope fence / barrier: lanes are free-running host threads here, so the barrier is a real one
inline void __builtin_amdgcn_fence(int, const char*) { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline void __builtin_amdgcn_wave_barrier() { hip_emu::st().waves[hip_emu::t_tid / 64].bar.arrive_and_wait(); }

// ---- gfx950 builtins the kernels use, restated for host threads ----------------------------------
// (so that the kernel sources carry ONE code path: the emulation lives here, not in #ifdef branches)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 4
template <typename T>
inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <typename T>
inline void __hip_atomic_store(T* p, T v, int, int) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
template <typename T>
inline bool __hip_atomic_compare_exchange_strong(T* p, T* expected, T desired, int, int, int)
{
    return __atomic_compare_exchange_n(p, expected, desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
}
inline float __hip_atomic_fetch_add(float* p, float x, int, int) { return emu_atomic_fadd(p, x); }
inline double __hip_atomic_fetch_add(double* p, double x, int, int) { return emu_atomic_fadd(p, x); }

// buffer resource + 16-byte buffer loads (the cache-policy immediate `aux` has no host meaning).  The RANGE CHECK is
// emulated: an out-of-range access returns zeros, as the hardware does -- a resource built too small shows up here.
struct __amdgpu_buffer_rsrc_t { const char* base; unsigned stride; unsigned num_records; };
typedef unsigned mi_u32x4 __attribute__((vector_size(16)));
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short stride, int num_records, int)
{
    return __amdgpu_buffer_rsrc_t{(const char*)p, (unsigned)(unsigned short)stride & 0x3fffu, (unsigned)num_records};
}
inline mi_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int)
{
    mi_u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned long long)voff + 16ull > (unsigned long long)r.num_records) return v;  // raw buffer: records are bytes
    std::memcpy(&v, r.base + voff + soff, 16);
    return v;
}
// matrix-core instructions: every lane publishes its A / B operand, then computes the accumulator
// registers it owns (fragment layouts: csrc/dense.hip header comment)
typedef float mi_f32x16 __attribute__((vector_size(64)));
typedef double mi_f64x4 __attribute__((vector_size(32)));
inline mi_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, mi_f32x16 acc, int, int, int)
{
    hip_emu::WaveState& w = hip_emu::st().waves[hip_emu::t_tid / 64];
    const int lane = hip_emu::t_tid % 64;
    float ab[2] = {a, b};
    std::memcpy(w.slot[lane], ab, sizeof(ab));
    w.bar.arrive_and_wait();
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;
        float s = acc[r];
        for (int k = 0; k < 2; ++k) {  // A[i][k] lives in lane i + 32 k, B[k][j] in lane j + 32 k
            float av[2], bv[2];
            std::memcpy(av, w.slot[i + 32 * k], sizeof(av));
            std::memcpy(bv, w.slot[j + 32 * k], sizeof(bv));
            s = __builtin_fmaf(av[0], bv[1], s);
        }
        acc[r] = s;
    }
    w.bar.arrive_and_wait();
    return acc;
}
inline mi_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, mi_f64x4 acc, int, int, int)
{
    hip_emu::WaveState& w = hip_emu::st().waves[hip_emu::t_tid / 64];
    const int lane = hip_emu::t_tid % 64;
    double ab[2] = {a, b};
    std::memcpy(w.slot[lane], ab, sizeof(ab));
    w.bar.arrive_and_wait();
    for (int r = 0; r < 4; ++r) {
        const int i = (lane >> 4) + 4 * r, j = lane & 15;
        double s = acc[r];
        for (int k = 0; k < 4; ++k) {  // A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k
            double av[2], bv[2];
            std::memcpy(av, w.slot[i + 16 * k], sizeof(av));
            std::memcpy(bv, w.slot[j + 16 * k], sizeof(bv));
            s = __builtin_fma(av[0], bv[1], s);
        }
        acc[r] = s;
    }
    w.bar.arrive_and_wait();
    return acc;
}

