// spmm_gather_probe.hip -- empirical ceiling of the SpMM gather on the headline matrix's OWN column stream (round 4).
// Not part of the library.  Generates the R-MAT 2^20 x 32 edge list on the host (same recursion as bench.py; the host RNG
// differs from torch's, the statistics are the same), sorts + de-duplicates it, and gathers rows of a 2^20 x 128 fp32 B in the
// order a CSR SpMM would, with the XCD-affine column slices of k_spmm, under
//   * every cache policy the buffer loads can carry on the COLD columns (aux: 1 = sc0, 2 = nt, 16 = sc1 and their sums),
//   * a hot set of H columns (H most referenced),
//   * ONE pass over all nonzeros, or TWO passes (hot nonzeros first, the rest afterwards: nothing cold can evict a hot row
//     while the hot pass runs), each pass with its own slice count.
// No A values, no C: the kernel only gathers (and sums, so the loads stay).  What it prints is the time the gather alone
// takes under each structure -- the number DESIGN.md's two-rate model only estimated.
//   build: hipcc --offload-arch=gfx950 -O3 spmm_gather_probe.hip -o spmm_gather_probe
//   run:   ./spmm_gather_probe [filter-substring]      (every dispatch is listed in order: PMC passes join on that)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int CH = 256;  // nonzeros per wave (k_spmm's chunk)

// One wave per chunk of CH column indices; LPN lanes x 16 B cover one slice of a B row; 64 / LPN lane groups x U loads in
// flight.  Block b runs on XCD b % 8; with S slices the XCDs form S sets, set s gathers bytes [s * 512 / S, (s + 1) * 512 / S)
// of each row.  Indices with bit 31 set are "cold" and use cache policy AUXC; the others use the default policy.
template <int LPN, int U, int AUXC>
__global__ void __launch_bounds__(256, 8) k_gather(const float* __restrict__ B, const int* __restrict__ idx, long nchunks,
                                                   long nidx, int slices, f4* __restrict__ out)
{
    constexpr int NG = 64 / LPN;
    const int lane = threadIdx.x % 64, wib = threadIdx.x / 64;
    long cb = blockIdx.x;
    int jlo = 0;
    if (slices > 1) {
        const int xcd = (int)(blockIdx.x & 7u), per = 8 / slices;
        cb = (long)(blockIdx.x >> 3) * per + (xcd % per);
        jlo = (xcd / per) * (128 / slices);
    }
    const long w = cb * 4 + wib;
    if (w >= nchunks) return;
    const int g = lane / LPN, li = lane % LPN;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0xffffffff, 0x00020000);
    const int* my = idx + w * CH;
    const long left = nidx - w * CH;
    const int len = left < CH ? (int)left : CH;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ncol = 128 / slices;  // floats per slice
    for (int j0 = 0; j0 < ncol; j0 += LPN * 4) {
        const int jc = jlo + j0 + li * 4;
        for (int p = g; p < len; p += NG * U) {
            int t[U];
            u4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int pp = p + u * NG;
                t[u] = my[pp < len ? pp : len - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned voff = ((unsigned)(t[u] & 0x7fffffff) * 128u + (unsigned)jc) * 4u;
                if (t[u] < 0) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, AUXC);
                else v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += __builtin_bit_cast(f4, v[u]);
        }
    }
    if (acc.x == 12345.678f) out[w * 64 + lane] = acc;  // never true: B holds small integers
}

static std::vector<int> g_cols;  // the CSR column stream
static std::vector<unsigned> g_cnt;

static void make_rmat(int scale, int per_row)
{
    const long n = 1l << scale, ne = n * per_row;
    std::vector<uint64_t> key(ne);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const double a = 0.57, b = 0.19, c = 0.19;
    for (long e = 0; e < ne; ++e) {
        uint64_t r = 0, cidx = 0;
        for (int l = 0; l < scale; ++l) {
            const double x = (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
            const int rb = x >= a + b, cbit = (x >= a && x < a + b) || x >= a + b + c;
            r = r * 2 + rb;
            cidx = cidx * 2 + cbit;
        }
        key[e] = r * (uint64_t)n + cidx;
    }
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    g_cols.resize(key.size());
    g_cnt.assign(n, 0);
    for (size_t i = 0; i < key.size(); ++i) {
        g_cols[i] = (int)(key[i] % (uint64_t)n);
        g_cnt[g_cols[i]]++;
    }
}

template <int LPN, int U>
static void launch(int aux, const float* B, const int* idx, long nidx, int slices, f4* out)
{
    const long nchunks = (nidx + CH - 1) / CH;
    unsigned grid = (unsigned)((nchunks + 3) / 4);
    if (slices > 1) grid = (unsigned)((grid + 8 / slices - 1) / (8 / slices)) * 8u;
#define L(A) hipLaunchKernelGGL((k_gather<LPN, U, A>), dim3(grid), dim3(256), 0, 0, B, idx, nchunks, nidx, slices, out)
    switch (aux) {
    case 0: L(0); break;
    case 1: L(1); break;
    case 2: L(2); break;
    case 3: L(3); break;
    case 16: L(16); break;
    case 17: L(17); break;
    case 18: L(18); break;
    case 19: L(19); break;
    default: printf("bad aux %d\n", aux); exit(1);
    }
#undef L
}

static void gather(int aux, const float* B, const int* idx, long nidx, int slices, f4* out)
{
    if (nidx == 0) return;
    const int lanes = 32 / slices;  // 16-byte lanes per slice of a 512-byte row
    if (lanes >= 32) launch<32, 4>(aux, B, idx, nidx, slices, out);
    else if (lanes == 16) launch<16, 4>(aux, B, idx, nidx, slices, out);
    else if (lanes == 8) launch<8, 4>(aux, B, idx, nidx, slices, out);
    else launch<4, 4>(aux, B, idx, nidx, slices, out);
}

int main(int argc, char** argv)
{
    const char* filter = argc > 1 ? argv[1] : "";
    const int scale = 20;
    const long n = 1l << scale;
    make_rmat(scale, 32);
    const long nnz = (long)g_cols.size();
    printf("R-MAT 2^%d: nnz %ld, gather volume %.2f GB per launch (512-byte rows)\n", scale, nnz, nnz * 512 / 1e9);
    std::vector<int> order(n);
    for (long i = 0; i < n; ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return g_cnt[x] > g_cnt[y]; });
    float* B;
    CK(hipMalloc(&B, n * 512));
    CK(hipMemset(B, 0, n * 512));
    int *d_all, *d_hot, *d_cold;
    CK(hipMalloc(&d_all, nnz * 4));
    CK(hipMalloc(&d_hot, nnz * 4));
    CK(hipMalloc(&d_cold, nnz * 4));
    f4* out;
    CK(hipMalloc(&out, (nnz / CH + 8) * 64 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int dispatch = 0;
    auto timed = [&](const std::string& name, auto&& fn, int kernels_per_rep) {
        if (*filter && name.find(filter) == std::string::npos) return;
        float best = 1e30f, sum = 0;
        const int reps = 5;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipEventRecord(e0));
            fn();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-64s min %.3f avg %.3f ms  [dispatches %d..%d, %d per rep]\n", name.c_str(), best, sum / reps, dispatch,
               dispatch + (reps + 1) * kernels_per_rep - 1, kernels_per_rep);
        dispatch += (reps + 1) * kernels_per_rep;
        fflush(stdout);
    };
    std::vector<int> h_all(nnz), h_hot, h_cold;
    std::vector<char> is_hot(n);
    for (int H : {0, 4096, 8192, 12288, 16384, 24576, 32768, 49152, 65536}) {
        std::fill(is_hot.begin(), is_hot.end(), 0);
        for (int k = 0; k < H; ++k) is_hot[order[k]] = 1;
        h_hot.clear();
        h_cold.clear();
        long hot_refs = 0;
        for (long i = 0; i < nnz; ++i) {
            const int c = g_cols[i];
            if (is_hot[c]) { h_all[i] = c; h_hot.push_back(c); ++hot_refs; }
            else { h_all[i] = (int)((unsigned)c | 0x80000000u); h_cold.push_back(h_all[i]); }
        }
        CK(hipMemcpy(d_all, h_all.data(), nnz * 4, hipMemcpyHostToDevice));
        if (!h_hot.empty()) CK(hipMemcpy(d_hot, h_hot.data(), h_hot.size() * 4, hipMemcpyHostToDevice));
        if (!h_cold.empty()) CK(hipMemcpy(d_cold, h_cold.data(), h_cold.size() * 4, hipMemcpyHostToDevice));
        const long nh = (long)h_hot.size(), nc = (long)h_cold.size();
        printf("--- H = %d hot columns: %.1f %% of the references\n", H, 100.0 * hot_refs / nnz);
        char nm[160];
        // one pass, cold policy sweep
        for (int S : {1, 2, 4}) {
            for (int aux : {0, 2, 1, 16, 17, 18, 3, 19}) {
                if (H == 0 && aux != 0 && aux != 2) continue;
                if (S != 2 && aux != 0 && aux != 2) continue;
                snprintf(nm, sizeof nm, "H=%d one-pass S=%d cold-aux=%d", H, S, aux);
                timed(nm, [&] { gather(aux, B, d_all, nnz, S, out); }, 1);
            }
        }
        if (H == 0) continue;
        // two passes, each alone and together
        for (int Sh : {2, 4, 8}) {
            snprintf(nm, sizeof nm, "H=%d hot-pass-alone S=%d", H, Sh);
            timed(nm, [&] { gather(0, B, d_hot, nh, Sh, out); }, 1);
        }
        for (int Sc : {1, 2}) {
            for (int aux : {0, 2}) {
                snprintf(nm, sizeof nm, "H=%d cold-pass-alone S=%d aux=%d", H, Sc, aux);
                timed(nm, [&] { gather(aux, B, d_cold, nc, Sc, out); }, 1);
            }
        }
        for (int Sh : {2, 4}) {
            for (int Sc : {1, 2}) {
                snprintf(nm, sizeof nm, "H=%d two-pass hot S=%d + cold S=%d aux=2", H, Sh, Sc);
                timed(nm, [&] { gather(0, B, d_hot, nh, Sh, out); gather(2, B, d_cold, nc, Sc, out); }, 2);
            }
        }
    }
    return 0;
}
