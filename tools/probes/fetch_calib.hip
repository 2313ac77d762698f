// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the access patterns of the gram / SpGEMM kernels?
// The guide calibrates it only for wide streaming reads (reports HALF the bytes on gfx950).  Three kernels read a KNOWN
// number of DRAM bytes from a 4 GiB array (far beyond L2 + Infinity Cache), each launched 3 times; run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib      (and TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum)
// and compare the counter with the printed byte counts:
//   k_stream   : 16 B per lane, fully coalesced            -> bytes touched = n * 16
//   k_seg32    : 8-lane groups read one random 32-byte segment (the sliced gram walk: 8 entries of a row's slice)
//                -> 32 B useful per segment; 64 B if DRAM is fetched in 64-B sectors, 128 B if whole lines are
//   k_word     : every lane reads one random 4-byte word (SpMV's x gather, SpGEMM's short slices)
// Build: hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_stream(const f4* __restrict__ a, long n, float* out)
{
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { f4 v = a[i]; s += v.x + v.w; }
    if (s == 123.456f) out[0] = s;
}
__device__ __forceinline__ unsigned long long mix(unsigned long long x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void __launch_bounds__(256) k_seg32(const float* __restrict__ a, long nseg_total, long nwords, float* out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long seg = t >> 3;
    const int sub = (int)(t & 7);
    float s = 0.f;
    for (long k = seg; k < nseg_total; k += ((long)gridDim.x * blockDim.x) >> 3) {
        const long base = (long)(mix((unsigned long long)k) % (unsigned long long)(nwords / 8)) * 8;  // 32-byte aligned segment
        s += a[base + sub];
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_word(const float* __restrict__ a, long n_total, long nwords, float* out)
{
    float s = 0.f;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n_total; k += (long)gridDim.x * blockDim.x)
        s += a[(long)(mix((unsigned long long)k * 7 + 1) % (unsigned long long)nwords)];
    if (s == 123.456f) out[0] = s;
}

int main()
{
    const long bytes = 4l << 30;
    float *a, *out;
    CK(hipMalloc(&a, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMalloc(&out, 64));
    const long nwords = bytes / 4;
    const long nseg = 1l << 26;   // 67 M segments of 32 B: 2.1 GB useful
    const long nword = 1l << 27;  // 134 M words
    for (int rep = 0; rep < 3; ++rep) {
        k_stream<<<4096, 256>>>((const f4*)a, bytes / 16, out);
        k_seg32<<<8192, 256>>>(a, nseg, nwords, out);
        k_word<<<8192, 256>>>(a, nword, nwords, out);
    }
    CK(hipDeviceSynchronize());
    printf("k_stream: %ld bytes read (16 B per lane, coalesced)\n", bytes);
    printf("k_seg32 : %ld segments: %ld useful bytes; %ld if fetched as 64-B sectors; %ld as 128-B lines\n", nseg, nseg * 32, nseg * 64, nseg * 128);
    printf("k_word  : %ld words: %ld useful bytes; %ld as 64-B sectors; %ld as 128-B lines\n", nword, nword * 4, nword * 64, nword * 128);
    return 0;
}
