// hotcold_probe.hip -- does marking COLD gathers non-temporal protect a HOT set in L2?
// Index stream: fraction `hot_frac` of the row reads goes to a hot set of H rows, the rest is
// uniform over W rows (512-byte rows).  Variants: plain loads for everything, or `nt` loads for the
// cold rows (selected per lane group by the sign bit of the staged index).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, int MODE>  // MODE 0: plain, 1: cold = nontemporal, 2: all nontemporal
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ B, const f4* __restrict__ Bcold,
                                                const int* __restrict__ idx, long per_wave, f4* __restrict__ out)
{
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x % 64, h = lane >> 5, li = lane & 31;
    const int* my = idx + wave * per_wave;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long k = 0; k < per_wave; k += 2 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = my[k + 2 * u + h];
            // two base pointers (equal at run time) keep the compiler from merging the two loads and
            // dropping the non-temporal hint
            const long off = (long)(t & 0x7fffffff) * 32 + li;
            if (MODE == 2 || (MODE == 1 && t < 0)) v[u] = __builtin_nontemporal_load(Bcold + off);
            else if (MODE == 1) v[u] = *(const volatile f4*)(B + off);
            else v[u] = B[off];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u]; }
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const long W = 1l << 20;  // 512 MB of rows
    f4 *B, *out;
    CK(hipMalloc(&B, W * 512));
    CK(hipMemset(B, 0, W * 512));
    const long nwaves = 256 * 32 * 4, per_wave = 512;
    const long nidx = nwaves * per_wave;
    int* idx;
    CK(hipMalloc(&idx, nidx * 4));
    CK(hipMalloc(&out, nwaves * 64 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<int> h(nidx);
    struct Cfg { long H; double frac; };
    for (Cfg c : {Cfg{0, 0.0}, Cfg{2048, 0.30}, Cfg{4096, 0.40}, Cfg{6144, 0.46}, Cfg{16384, 0.60}, Cfg{32768, 0.70}}) {
        unsigned long long s = 88172645463325252ull;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
        for (long i = 0; i < nidx; ++i) {
            const bool hot = c.H > 0 && (double)(rnd() % 1000000) / 1e6 < c.frac;
            // hot rows are spread over the array (stride) so they are not one contiguous block
            h[i] = hot ? (int)((rnd() % c.H) * (W / (c.H ? c.H : 1))) : (int)((rnd() % W) | 0x80000000u);
        }
        CK(hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 3; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k_gather<4, 0><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else if (mode == 1) k_gather<4, 1><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else k_gather<4, 2><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("hot set %6ld rows (%5.1f MB) hot fraction %.2f mode %s: %.3f ms -> %.1f GB/s\n", c.H, c.H * 512 / 1048576.0,
                   c.frac, mode == 0 ? "plain   " : mode == 1 ? "cold=nt " : "all=nt  ", ms, (double)nidx * 512 / ms / 1e6);
        }
    }
    return 0;
}
