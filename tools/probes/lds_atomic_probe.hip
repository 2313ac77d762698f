// lds_atomic_probe.hip -- what does an LDS floating-point atomic cost on gfx950, and what replaces it?  (round 4: the dense gram
// kernel issues ~9.7 k of them per 152 KiB tile and its knock-out build without them was 23 ms faster)
// One 1024-thread workgroup per CU, a 152 KiB tile, every lane adds to pseudo-random cells (an LCG per lane: no loads), ITER
// batches of 8 wave-instructions per wave.  FILL % of the cells are non-zero before the first add (a compare-and-swap against zero
// fails there).
//   add_f32 / add_f64    ds_add_f32 / ds_add_f64 (no return)           add_u32 / add_u64   the integer atomics
//   add_rtn_f32          with return                                     rmw                 read + add + write (UNSAFE: rate only)
//   swap+fadd            compare-and-swap against zero, float atomic where the cell was taken (8 swaps in flight)
//   swap+cas+fadd        ... one compare-and-swap retry on the returned value, float atomic only after that fails too
//   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_probe.hip -o lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T> struct W { using u = unsigned; };
template <> struct W<double> { using u = unsigned long long; };
__device__ inline unsigned bits(float x) { return __float_as_uint(x); }
__device__ inline unsigned long long bits(double x) { return (unsigned long long)__double_as_longlong(x); }
__device__ inline float val(unsigned b) { return __uint_as_float(b); }
__device__ inline double val(unsigned long long b) { return __longlong_as_double((long long)b); }

enum { ADD_F = 0, ADD_U = 1, ADD_RTN = 2, RMW = 3, SWAP_FADD = 4, SWAP_CAS_FADD = 5 };

template <typename T, int MODE, int ACTIVE_PCT, int FILL>
__global__ void __launch_bounds__(1024) k_probe(int iters, T* out)
{
    constexpr int TILE = 155648 / sizeof(T);
    using U = typename W<T>::u;
    __shared__ T acc[TILE];
    for (int k = threadIdx.x; k < TILE; k += 1024) acc[k] = ((k * 7919) % 100) < FILL ? (T)1 : (T)0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const bool active = (int)((threadIdx.x * 37u) % 100u) < ACTIVE_PCT;
    T sink = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            idx[u] = (s >> 8) % TILE;
        }
        if (MODE == SWAP_FADD || MODE == SWAP_CAS_FADD) {
            U was[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                was[u] = 0;
                if (active) was[u] = atomicCAS(reinterpret_cast<U*>(&acc[idx[u]]), (U)0, bits((T)1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(was[u]));
            if (MODE == SWAP_CAS_FADD) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (was[u] != 0) {
                        const U e = was[u];
                        was[u] = atomicCAS(reinterpret_cast<U*>(&acc[idx[u]]), e, bits(val(e) + (T)1)) == e ? (U)0 : (U)1;
                    }
#pragma unroll
                for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(was[u]));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (was[u] != 0) atomicAdd(&acc[idx[u]], (T)1);
            continue;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!active) continue;
            if (MODE == ADD_F) atomicAdd(&acc[idx[u]], (T)1);
            else if (MODE == ADD_U) atomicAdd(reinterpret_cast<U*>(&acc[idx[u]]), (U)1);
            else if (MODE == ADD_RTN) sink += atomicAdd(&acc[idx[u]], (T)1);
            else acc[idx[u]] += (T)1;
        }
    }
    __syncthreads();
    T t = sink;
    for (int k = threadIdx.x; k < TILE; k += 1024) t += acc[k];
    if (t == (T)12345.678) out[blockIdx.x] = t;
}

template <typename T, int MODE, int PCT, int FILL>
static void run(const char* name, void* out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 500;
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<T, MODE, PCT, FILL>), dim3(256), dim3(1024), 0, 0, iters, (T*)out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double wave_instr = 16.0 * iters * 8;  // accumulate operations per CU, in wave instructions
    const double lanes = wave_instr * 64 * PCT / 100.0;
    printf("%-4s %-16s %3d %% lanes active, %3d %% of the cells taken: %8.3f ms -> %6.1f cycles per wave-level accumulate, %5.2f per active lane\n",
           sizeof(T) == 4 ? "f32" : "f64", name, PCT, FILL, best, best * 1e-3 * 2.4e9 / wave_instr, best * 1e-3 * 2.4e9 / lanes);
    fflush(stdout);
}

int main()
{
    void* out;
    CK(hipMalloc(&out, 8192));
    run<float, ADD_F, 100, 0>("add_f32", out);
    run<float, ADD_F, 60, 0>("add_f32", out);
    run<float, ADD_F, 10, 0>("add_f32", out);
    run<float, ADD_U, 100, 0>("add_u32", out);
    run<float, ADD_RTN, 100, 0>("add_rtn_f32", out);
    run<float, RMW, 100, 0>("rmw (unsafe)", out);
    run<float, SWAP_FADD, 100, 0>("swap+fadd", out);
    run<float, SWAP_FADD, 60, 0>("swap+fadd", out);
    run<float, SWAP_FADD, 60, 10>("swap+fadd", out);
    run<float, SWAP_FADD, 60, 50>("swap+fadd", out);
    run<float, SWAP_FADD, 100, 100>("swap+fadd", out);
    run<float, SWAP_CAS_FADD, 60, 10>("swap+cas+fadd", out);
    run<float, SWAP_CAS_FADD, 60, 50>("swap+cas+fadd", out);
    run<float, SWAP_CAS_FADD, 100, 100>("swap+cas+fadd", out);
    run<double, ADD_F, 100, 0>("add_f64", out);
    run<double, ADD_F, 60, 0>("add_f64", out);
    run<double, ADD_U, 100, 0>("add_u64", out);
    run<double, RMW, 100, 0>("rmw (unsafe)", out);
    run<double, SWAP_FADD, 100, 0>("swap+fadd", out);
    run<double, SWAP_FADD, 100, 50>("swap+fadd", out);
    run<double, SWAP_FADD, 100, 100>("swap+fadd", out);
    run<double, SWAP_CAS_FADD, 100, 50>("swap+cas+fadd", out);
    run<double, SWAP_CAS_FADD, 100, 100>("swap+cas+fadd", out);
    return 0;
}
