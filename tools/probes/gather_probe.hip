// gather_probe.hip -- calibration micro-benchmark (not part of the library).
// Measures (1) streaming copy bandwidth (the "measured HBM roofline" SURVEY section 8d asks for) and
// (2) the bandwidth of the SpMM access pattern in isolation: half-waves reading random 512-byte
// rows (32 lanes x 16 B) out of a working set of W rows, U loads in flight per lane, 32 waves/CU.
// Build: hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o gather_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ B, const int* __restrict__ idx, long per_wave,
                                                float4* __restrict__ out)
{
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x % 64, h = lane >> 5, li = lane & 31;
    const int* my = idx + wave * per_wave;
    float4 acc = {0, 0, 0, 0};
    for (long k = 0; k < per_wave; k += 2 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = B[(long)my[k + 2 * u + h] * 32 + li];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

int main()
{
    const long MAXROWS = 4l << 20;  // 2 GiB of 512-B rows
    float4 *B, *out;
    CK(hipMalloc(&B, MAXROWS * 512));
    CK(hipMemset(B, 0, MAXROWS * 512));
    const long nwaves = 256 * 32 * 4, per_wave = 512;  // 16.8 M row reads = 8.6 GB per launch
    const long nidx = nwaves * per_wave;
    int* idx;
    CK(hipMalloc(&idx, nidx * 4));
    CK(hipMalloc(&out, nwaves * 64 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // streaming copy
    {
        float4* dst;
        const long n = (1l << 30) / 16;  // 1 GiB
        CK(hipMalloc(&dst, n * 16));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            k_copy<<<2048, 256>>>(B, dst, n);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB: %.3f ms -> %.1f GB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e6);
        CK(hipFree(dst));
    }
    std::vector<int> h(nidx);
    for (long W : {4096l, 8192l, 32768l, 131072l, 262144l, 524288l, 1048576l, 4194304l}) {
        unsigned long long s = 88172645463325252ull;
        for (long i = 0; i < nidx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (unsigned long long)W); }
        CK(hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
        for (int U : {4, 8}) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (U == 4) k_gather<4><<<nwaves / 4, 256>>>(B, idx, per_wave, out);
                else k_gather<8><<<nwaves / 4, 256>>>(B, idx, per_wave, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("gather W=%8ld rows (%7.1f MB) U=%d: %.3f ms -> %.1f GB/s\n", W, W * 512 / 1048576.0, U, ms,
                   (double)nidx * 512 / ms / 1e6);
        }
    }
    return 0;
}
