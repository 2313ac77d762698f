// lds_atomic_probe.hip -- what does an LDS float atomic cost on gfx950?  (round 4: the dense gram kernel issues ~9.7 k of them per
// 152 KiB tile and its knock-out build without them was 23 ms faster)
// One 1024-thread workgroup per CU, a 152 KiB tile, every lane adds to pseudo-random cells (an LCG per lane: no loads), ITER
// wave-instructions per wave.  Variants: float atomic add (ds_add_f32), integer atomic add (ds_add_u32), float add with return
// (ds_add_rtn_f32), plain read-modify-write (ds_read + add + ds_write: NOT safe, the rate only), with all / 60 % of the lanes active.
//   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_probe.hip -o lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int TILE = 38912;

template <int MODE, int ACTIVE_PCT>
__global__ void __launch_bounds__(1024) k_probe(int iters, float* out)
{
    __shared__ float acc[TILE];
    for (int k = threadIdx.x; k < TILE; k += 1024) acc[k] = 0.f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const bool active = (int)((threadIdx.x * 37u) % 100u) < ACTIVE_PCT;
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned idx[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            idx[u] = (s >> 8) % TILE;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!active) continue;
            if (MODE == 0) atomicAdd(&acc[idx[u]], 1.0f);
            else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(&acc[idx[u]]), 1u);
            else if (MODE == 2) sink += atomicAdd(&acc[idx[u]], 1.0f);
            else acc[idx[u]] += 1.0f;
        }
    }
    __syncthreads();
    float t = sink;
    for (int k = threadIdx.x; k < TILE; k += 1024) t += acc[k];
    if (t == 12345.678f) out[blockIdx.x] = t;
}

template <int MODE, int PCT>
static void run(const char* name, float* out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    float ms = 0, best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<MODE, PCT>), dim3(256), dim3(1024), 0, 0, iters, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double wave_instr = 16.0 * iters * 8;  // per CU
    const double lanes = wave_instr * 64 * PCT / 100.0;
    printf("%-28s %3d %% lanes: %8.3f ms  -> %6.1f cycles per wave-instruction, %5.2f cycles per active lane (2.4 GHz, per CU)\n", name,
           PCT, best, best * 1e-3 * 2.4e9 / wave_instr, best * 1e-3 * 2.4e9 / lanes);
}

int main()
{
    float* out;
    CK(hipMalloc(&out, 4096));
    run<0, 100>("ds_add_f32 (no return)", out);
    run<0, 60>("ds_add_f32 (no return)", out);
    run<0, 25>("ds_add_f32 (no return)", out);
    run<1, 100>("ds_add_u32 (no return)", out);
    run<1, 60>("ds_add_u32 (no return)", out);
    run<2, 100>("ds_add_rtn_f32", out);
    run<3, 100>("read + add + write (unsafe)", out);
    run<3, 60>("read + add + write (unsafe)", out);
    return 0;
}
