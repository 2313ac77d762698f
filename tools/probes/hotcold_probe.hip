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

__device__ __forceinline__ f4 load_nt_asm(const f4* p)
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p));
    return v;
}

template <int U, int MODE>  // MODE 0: plain, 1: cold = nt (hot = volatile), 2: all nt, 3: hot plain + cold nt (asm)
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ B, const f4* __restrict__ Bcold,
                                                const int* __restrict__ idx, long per_wave, f4* __restrict__ out)
{
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    const int lane = threadIdx.x % 64, h = lane >> 5, li = lane & 31;
    const int* my = idx + wave * per_wave;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    // raw buffer resource over all of B (< 4 GiB): base, stride 0, num_records in bytes, gfx9 raw flags
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    for (long k = 0; k < per_wave; k += 2 * U) {
        f4 v[U];
        int tt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) tt[u] = my[k + 2 * u + h];
        if (MODE == 3) {  // indices must be in registers before the hand-waited loads are issued
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tt[u];
            // two base pointers (equal at run time) keep the compiler from merging the two loads and
            // dropping the non-temporal hint
            const long off = (long)(t & 0x7fffffff) * 32 + li;
            if (MODE == 4) {
                // buffer loads: cache policy is an immediate of the instruction (aux 2 = nt)
                const unsigned voff = (unsigned)(off * 16);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                u4 r;
                if (t < 0) r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2);
                else r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
                v[u] = __builtin_bit_cast(f4, r);
            } else if (MODE == 3) {
                if (t < 0) v[u] = load_nt_asm(Bcold + off);
                else v[u] = B[off];
            } else if (MODE == 2 || (MODE == 1 && t < 0)) v[u] = __builtin_nontemporal_load(Bcold + off);
            else if (MODE == 1) v[u] = *(const volatile f4*)(B + off);
            else v[u] = B[off];
        }
        if (MODE == 3) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
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
        for (int mode = 0; mode < 5; ++mode) {
            float ms = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k_gather<4, 0><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else if (mode == 1) k_gather<4, 1><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else if (mode == 2) k_gather<4, 2><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else if (mode == 3) k_gather<4, 3><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                else k_gather<4, 4><<<nwaves / 4, 256>>>(B, B, idx, per_wave, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("hot set %6ld rows (%5.1f MB) hot fraction %.2f mode %s: %.3f ms -> %.1f GB/s\n", c.H, c.H * 512 / 1048576.0,
                   c.frac, mode == 0 ? "plain   " : mode == 1 ? "cold=nt " : mode == 2 ? "all=nt  " : mode == 3 ? "hot=plain cold=nt(asm)" : "buffer loads hot=0 cold=nt", ms, (double)nidx * 512 / ms / 1e6);
        }
    }
    return 0;
}
