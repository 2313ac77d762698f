// alloc_probe.hip -- what does a fresh multi-GB device allocation cost, and does the virtual-memory API do better?
// (a one-shot mi_sparse_spmm of the literal configs[2] allocates 39 GB + 78 GB of result arrays: 2.6 s first call in round 2)
// Build: hipcc --offload-arch=gfx950 -O2 alloc_probe.hip -o alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define printf(...) (printf(__VA_ARGS__), fflush(stdout))
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n, size_t stride) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; if (i < n) p[i] = 1; }

int main()
{
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (double gb : {1.0, 8.0, 39.0, 78.0}) {
        const size_t n = (size_t)(gb * (1ull << 30));
        void* p = nullptr;
        double t0 = now();
        CK(hipMalloc(&p, n));
        double t1 = now();
        touch<<<(unsigned)((n / (2u << 20)) / 256 + 1), 256>>>((char*)p, n, 2u << 20);  // one byte per 2 MiB page
        CK(hipDeviceSynchronize());
        double t2 = now();
        CK(hipFree(p));
        double t3 = now();
        printf("hipMalloc %5.1f GiB: alloc %8.2f ms, first touch %7.2f ms, free %8.2f ms\n", gb, t1 - t0, t2 - t1, t3 - t2);
        // again (does the driver keep anything?)
        t0 = now();
        CK(hipMalloc(&p, n));
        t1 = now();
        CK(hipFree(p));
        printf("          again      : alloc %8.2f ms, free %8.2f ms\n", t1 - t0, now() - t1);
    }
    // virtual memory API: reserve once, create + map physical chunks
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity %zu bytes\n", gran);
    for (double gb : {8.0, 78.0}) {
        const size_t n = ((size_t)(gb * (1ull << 30)) + gran - 1) / gran * gran;
        for (size_t chunk : {(size_t)1 << 30, n}) {
            void* va = nullptr;
            double t0 = now();
            CK(hipMemAddressReserve(&va, n, 0, nullptr, 0));
            double t1 = now();
            std::vector<hipMemGenericAllocationHandle_t> hs;
            for (size_t off = 0; off < n; off += chunk) {
                const size_t sz = off + chunk <= n ? chunk : n - off;
                hipMemGenericAllocationHandle_t h;
                CK(hipMemCreate(&h, sz, &prop, 0));
                CK(hipMemMap((char*)va + off, sz, 0, h, 0));
                hs.push_back(h);
            }
            double t2 = now();
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, n, &acc, 1));
            double t3 = now();
            touch<<<(unsigned)((n / (2u << 20)) / 256 + 1), 256>>>((char*)va, n, 2u << 20);
            CK(hipDeviceSynchronize());
            double t4 = now();
            CK(hipMemUnmap(va, n));
            for (auto h : hs) CK(hipMemRelease(h));
            CK(hipMemAddressFree(va, n));
            double t5 = now();
            printf("VMM %5.1f GiB in %zu chunk(s): reserve %.2f ms, create+map %.2f ms, set access %.2f ms, touch %.2f ms, teardown %.2f ms\n",
                   gb, hs.size(), t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
        }
    }
    // stream-ordered allocator
    {
        hipMemPool_t pool;
        CK(hipDeviceGetDefaultMemPool(&pool, 0));
        uint64_t thr = ~0ull;
        CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
        for (int rep = 0; rep < 2; ++rep) {
            const size_t n = (size_t)78 << 30;
            void* p = nullptr;
            double t0 = now();
            CK(hipMallocAsync(&p, n, 0));
            CK(hipStreamSynchronize(0));
            double t1 = now();
            CK(hipFreeAsync(p, 0));
            CK(hipStreamSynchronize(0));
            printf("hipMallocAsync 78 GiB (rep %d): alloc %.2f ms, free %.2f ms\n", rep, t1 - t0, now() - t1);
        }
    }
    return 0;
}
