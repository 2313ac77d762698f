// global_atomic_probe.hip -- round 4: what does a floating-point atomic to GLOBAL memory cost on gfx950 when every region is
// owned by one workgroup?  (Question behind it: can the numeric big-row SpGEMM add every product straight into cval[row
// start + rank of the column] -- rank from an LDS bitmap -- instead of going through a hash table and a slice table?)
// Each 256-thread workgroup owns REGION consecutive cells of a large buffer per item and walks `items` regions; every lane issues
// 8 atomics per batch, ITER batches per item.
//   pattern random    : uniformly random cell of the region per lane
//   pattern clustered : a wave's 64 lanes fall in a window of 64 * SPREAD consecutive cells (sorted entries of one row of B map to
//                       increasing ranks), i.e. ~16 / SPREAD lanes per 128-byte line of doubles
//   scope wg / agent  : __HIP_MEMORY_SCOPE_WORKGROUP (may be performed in the XCD's own L2) / _AGENT
//   zero              : the workgroup first stores zeros over its region (lines become resident in its L2) / not (lines come from HBM)
//   build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics global_atomic_probe.hip -o global_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T, int SCOPE, int CLUSTER, int ZERO>
__global__ void __launch_bounds__(256) k_probe(T* buf, long long region, int items, int iters, int spread)
{
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < items; ++it) {
        T* r = buf + ((long long)blockIdx.x * items + it) * region;
        if (ZERO) {
            for (long long k = threadIdx.x; k < region; k += 256) r[k] = (T)0;
            __syncthreads();
        }
        for (int b = 0; b < iters; ++b) {
            long long idx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s = s * 1664525u + 1013904223u;
                if (CLUSTER) {
                    // wave-uniform window start (derived from a wave-uniform LCG), lane offset with jitter inside its own stripe
                    unsigned ws = (unsigned)((b * 8 + u) * 2654435761u + (threadIdx.x >> 6) * 97u + blockIdx.x * 31u + it * 7u);
                    const long long span = 64ll * spread;
                    const long long w0 = (long long)(ws % (unsigned)(region - span + 1));
                    idx[u] = w0 + (long long)lane * spread + (long long)((s >> 8) % (unsigned)spread);
                } else {
                    idx[u] = (long long)((s >> 8) % (unsigned)region);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                __hip_atomic_fetch_add(&r[idx[u]], (T)1, __ATOMIC_RELAXED,
                                       SCOPE ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (ZERO) __syncthreads();
    }
}

template <typename T, int SCOPE, int CLUSTER, int ZERO>
static void run(const char* name, T* buf, long long cells, long long region, int iters, int spread)
{
    const int blocks = 256 * 8;
    int items = (int)(cells / region / blocks);
    if (items < 1) { printf("%-44s buffer too small\n", name); return; }
    if (items > 64) items = 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(buf, 0, sizeof(T) * (size_t)cells));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_probe<T, SCOPE, CLUSTER, ZERO>), dim3(blocks), dim3(256), 0, 0, buf, region, items, iters, spread);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double n = (double)blocks * items * iters * 8.0 * 256.0;
    // spot check: the sum of one region equals the atomics issued into it
    printf("%-44s region %8lld cells, %2d items x %4d batches: %8.3f ms  %8.1f G atomics/s  (%.1f per cell)\n", name, region, items,
           iters, best, n / best / 1e6, (double)iters * 8 * 256 / (double)region);
}

int main()
{
    const long long bytes = 8ll << 30;  // 8 GiB buffer
    void* buf;
    CK(hipMalloc(&buf, (size_t)bytes));
    for (long long region : {8192ll, 131072ll}) {
        const int iters = (int)(region * 2 / (8 * 256));  // ~2 atomics per cell, as the product (2.15 products per entry of C)
        double* d = (double*)buf;
        float* f = (float*)buf;
        const long long cd = bytes / 8, cf = bytes / 4;
        run<double, 0, 0, 0>("f64 wg-scope   random", d, cd, region, iters, 1);
        run<double, 1, 0, 0>("f64 agent      random", d, cd, region, iters, 1);
        run<double, 0, 1, 0>("f64 wg-scope   clustered x2", d, cd, region, iters, 2);
        run<double, 1, 1, 0>("f64 agent      clustered x2", d, cd, region, iters, 2);
        run<double, 0, 1, 0>("f64 wg-scope   clustered x8", d, cd, region, iters, 8);
        run<double, 0, 1, 1>("f64 wg-scope   clustered x2, zeroed first", d, cd, region, iters, 2);
        run<double, 0, 0, 1>("f64 wg-scope   random, zeroed first", d, cd, region, iters, 1);
        run<float, 0, 1, 0>("f32 wg-scope   clustered x2", f, cf, region, iters, 2);
        run<float, 1, 1, 0>("f32 agent      clustered x2", f, cf, region, iters, 2);
        run<float, 0, 0, 0>("f32 wg-scope   random", f, cf, region, iters, 1);
    }
    return 0;
}
