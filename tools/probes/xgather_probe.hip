// xgather_probe.hip -- calibration micro-benchmark for SpMV (not part of the library).
// What bounds y = A x for ~32 nonzeros per row and a 4 MB x?  Every nonzero needs ONE 4-byte element of x at a random
// address: one L2 request per lane unless the line is already in the CU's L1.  The probe isolates that access pattern:
//   stream      : the (col, val) stream alone (8 B per nonzero, 16-byte loads)
//   gather      : stream + x[col] from global memory (uniform / R-MAT-skewed columns)
//   gather+lds  : the H most popular columns of the skewed stream are served from an LDS copy of their x entries
//                 (tag bit 31 + rank in the column word), the rest from global memory
// Output: ms and "GB/s algorithmic" = nnz * 8 B / t  (the figure SpMV's roofline fraction is quoted in).
// Build: hipcc --offload-arch=gfx950 -O3 xgather_probe.hip -o xgather_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef int i4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: stream only; 1: + global gather; 2: + gather with an LDS-resident hot set of H entries
template <int MODE, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_probe(const i4* __restrict__ col, const f4* __restrict__ val,
                                                      const float* __restrict__ x, const float* __restrict__ xhot, int H,
                                                      long n4_per_wave, float* __restrict__ out)
{
    extern __shared__ float s_hot[];
    if (MODE == 2) {
        for (int k = threadIdx.x; k < H / 4; k += blockDim.x) reinterpret_cast<f4*>(s_hot)[k] = reinterpret_cast<const f4*>(xhot)[k];
        __syncthreads();
    }
    const long wave = (long)blockIdx.x * WAVES + threadIdx.x / 64;
    const int lane = threadIdx.x % 64;
    const i4* c = col + wave * n4_per_wave;
    const f4* v = val + wave * n4_per_wave;
    float acc = 0.f;
    for (long k = lane; k < n4_per_wave; k += 2 * 64) {  // two 16-byte (col, val) pairs per lane in flight = 8 nonzeros
        i4 cc[2];
        f4 vv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long kk = k + u * 64 < n4_per_wave ? k + u * 64 : k;
            cc[u] = __builtin_nontemporal_load(c + kk);
            vv[u] = __builtin_nontemporal_load(v + kk);
        }
        float xv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = cc[u][e];
                if (MODE == 0) xv[u][e] = (float)(ci & 1);
                else if (MODE == 1) xv[u][e] = x[ci & 0x7fffffff];
                else xv[u][e] = ci < 0 ? s_hot[ci & 0x7fffffff] : x[ci];
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_fmaf(vv[u][e], xv[u][e], acc);
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    const long ncols = 1l << 20;
    const long nwaves = 256l * 32 * 4;       // 32768 waves
    const long n4_per_wave = 256;            // 1024 nonzeros per wave -> 33.5 M nonzeros
    const long nnz = nwaves * n4_per_wave * 4;
    int *col, *colh;
    float *val, *x, *xhot, *out;
    CK(hipMalloc(&col, nnz * 4));
    CK(hipMalloc(&colh, nnz * 4));
    CK(hipMalloc(&val, nnz * 4));
    CK(hipMalloc(&x, ncols * 4));
    CK(hipMalloc(&xhot, 65536 * 4));
    CK(hipMalloc(&out, nwaves * 64 * 4));
    CK(hipMemset(val, 0, nnz * 4));
    CK(hipMemset(x, 0, ncols * 4));
    CK(hipMemset(xhot, 0, 65536 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<int> h(nnz), hh(nnz);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    // popularity rank of a column under the R-MAT column marginal: P(bit = 1) = 0.24 per level -> fewer one-bits = hotter
    std::vector<int> order(ncols), rank(ncols);
    for (long i = 0; i < ncols; ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [](int a, int b) { return __builtin_popcount(a) < __builtin_popcount(b); });
    for (long i = 0; i < ncols; ++i) rank[order[i]] = (int)i;
    for (int skew = 0; skew < 2; ++skew) {
        for (long i = 0; i < nnz; ++i) {
            int c = 0;
            if (!skew) c = (int)(rnd() % ncols);
            else {
                const unsigned long long r = rnd();
                for (int b = 0; b < 20; ++b) c = (c << 1) | (((r >> (3 * b)) & 7) < 2 ? 1 : 0);  // P(1) = 0.25
            }
            h[i] = c;
        }
        CK(hipMemcpy(col, h.data(), nnz * 4, hipMemcpyHostToDevice));
        auto run = [&](const char* name, int mode, int waves, int H, const int* cols) {
            float ms = 0, best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                const dim3 grid((unsigned)(nwaves / waves)), block(waves * 64);
                const size_t lds = mode == 2 ? (size_t)H * 4 : 0;
                if (mode == 0) k_probe<0, 4><<<grid, block, 0>>>((const i4*)cols, (const f4*)val, x, xhot, H, n4_per_wave, out);
                else if (mode == 1) k_probe<1, 4><<<grid, block, 0>>>((const i4*)cols, (const f4*)val, x, xhot, H, n4_per_wave, out);
                else if (waves == 4) k_probe<2, 4><<<grid, block, lds>>>((const i4*)cols, (const f4*)val, x, xhot, H, n4_per_wave, out);
                else k_probe<2, 8><<<grid, block, lds>>>((const i4*)cols, (const f4*)val, x, xhot, H, n4_per_wave, out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%-8s %-34s: %.3f ms -> %.1f GB/s algorithmic (8 B per nonzero)\n", skew ? "skewed" : "uniform", name, best,
                   (double)nnz * 8 / best / 1e6);
        };
        run("stream only", 0, 4, 0, col);
        run("stream + global gather", 1, 4, 0, col);
        if (skew) {
            for (int H : {4096, 8192, 16384, 32768}) {
                long hot = 0;
                for (long i = 0; i < nnz; ++i) {
                    const bool is_hot = rank[h[i]] < H;
                    hot += is_hot;
                    hh[i] = is_hot ? (int)(0x80000000u | (unsigned)rank[h[i]]) : h[i];
                }
                CK(hipMemcpy(colh, hh.data(), nnz * 4, hipMemcpyHostToDevice));
                char name[96];
                for (int waves : {4, 8}) {
                    snprintf(name, sizeof name, "LDS hot set H=%5d (%4.1f%% refs) %d waves", H, 100.0 * hot / nnz, waves);
                    run(name, 2, waves, H, colh);
                }
            }
        }
    }
    return 0;
}
