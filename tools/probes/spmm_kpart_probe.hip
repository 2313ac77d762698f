// spmm_kpart_probe.hip -- round 5: does a COLUMN-PARTITIONED gather (each XCD only ever touches its own 1/P of the rows
// of B, so the eight 4 MB L2s hold P times more DISTINCT rows between them) beat the row-owned gather of k_spmm on the
// headline matrix's own column stream?  Gather only (no A values, no C), same conventions as spmm_gather_probe.hip.
//
// The headline R-MAT's nonzeros sit in LONG rows (the 6 % of the rows with >= 64 nonzeros hold ~80 % of them), so the price of
// a column partition -- one partial output row per (row, partition) -- is small for exactly the rows that carry the gather:
//   long rows  (>= T nonzeros): nonzeros split by part(col) in [0, P); partition p is gathered by XCD set p only
//   short rows: gathered as k_spmm does today (row-owned, S = 2 slices, cold columns non-temporal)
// Printed: time of the long pass, the short pass, and the bytes of partials the real kernel would add.
//   build: hipcc --offload-arch=gfx950 -O3 spmm_kpart_probe.hip -o spmm_kpart_probe
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

constexpr int CH = 256;

struct Parts { long cs[33]; long nblk; };  // chunk range of partition p: [cs[p], cs[p + 1]); PART = 2: workgroups per XCD and phase

// PART = 0: k_spmm's mapping (S column slices, every XCD sees every chunk of its slice set)
// PART = 1: P = 8 / slices partitions; XCD x = b & 7 -> partition x % P, slice x / P; chunk = cs[p] + (b >> 3) * 4 + wave
template <int LPN, int U, int AUXC, int PART>
__global__ void __launch_bounds__(256, 8) k_gather(const float* __restrict__ B, const int* __restrict__ idx, long nchunks,
                                                   long nidx, int slices, Parts parts, f4* __restrict__ out)
{
    constexpr int NG = 64 / LPN;
    const int lane = threadIdx.x % 64, wib = threadIdx.x / 64;
    long w;
    int jlo = 0;
    const int xcd = (int)(blockIdx.x & 7u);
    if (PART == 2) {  // 8 * phases partitions: XCD x takes partitions x, x + 8, ... one after the other (blocks are dispatched in order)
        const long bi = (long)(blockIdx.x >> 3);
        const int p = xcd + 8 * (int)(bi / parts.nblk);
        w = parts.cs[p] + (bi % parts.nblk) * 4 + wib;
        if (w >= parts.cs[p + 1]) return;
    } else if (PART) {
        const int P = 8 / slices, p = xcd % P;
        jlo = (xcd / P) * (128 / slices);
        w = parts.cs[p] + (long)(blockIdx.x >> 3) * 4 + wib;
        if (w >= parts.cs[p + 1]) return;
    } else {
        long cb = blockIdx.x;
        if (slices > 1) {
            const int per = 8 / slices;
            cb = (long)(blockIdx.x >> 3) * per + (xcd % per);
            jlo = (xcd / per) * (128 / slices);
        }
        w = cb * 4 + wib;
        if (w >= nchunks) return;
    }
    const int g = lane / LPN, li = lane % LPN;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0xffffffff, 0x00020000);
    const int* my = idx + w * CH;
    const long left = nidx - w * CH;
    const int len = left < CH ? (int)left : CH;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ncol = 128 / slices;
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
    if (acc.x == 12345.678f) out[w * 64 + lane] = acc;
}

// Persistent 1024-thread workgroups (one per CU), the H most referenced rows of B of the workgroup's partition copied into LDS
// once: an index with bit 30 set is (bit 30 | slot) and is served from that copy -- it never reaches the L2.
template <int U>
__global__ void __launch_bounds__(1024) k_gather_lds(const float* __restrict__ B, const int* __restrict__ idx, Parts parts, long nidx,
                                                     const int* __restrict__ hot_list, int H, f4* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f4* cache = reinterpret_cast<f4*>(smem);  // [H][32] pieces of 16 bytes
    const int xcd = (int)(blockIdx.x & 7u), p = xcd;
    const int tid = threadIdx.x, lane = tid % 64, wave = tid / 64;
    for (int i = tid; i < H * 32; i += 1024) {
        const int row = hot_list[p * H + i / 32];
        cache[i] = *reinterpret_cast<const f4*>(B + (long)row * 128 + (i % 32) * 4);
    }
    __syncthreads();
    const int g = lane / 32, li = lane % 32;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0xffffffff, 0x00020000);
    const long wgx = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long w = parts.cs[p] + wgx * 16 + wave; w < parts.cs[p + 1]; w += nwg * 16) {
        const int* my = idx + w * CH;
        const long left = nidx - w * CH;
        const int len = left < CH ? (int)left : CH;
        for (int q = g; q < len; q += 2 * U) {
            int t[U];
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = my[q + 2 * u < len ? q + 2 * u : len - 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (t[u] & 0x40000000) v[u] = cache[(t[u] & 0xffff) * 32 + li];
                else v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((unsigned)t[u] * 128u + (unsigned)li * 4u) * 4u, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    if (acc.x == 12345.678f) out[blockIdx.x * 1024 + tid] = acc;
}

static std::vector<int> g_cols;
static std::vector<long> g_ptr;
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
    g_ptr.assign(n + 1, 0);
    for (size_t i = 0; i < key.size(); ++i) {
        g_cols[i] = (int)(key[i] % (uint64_t)n);
        g_cnt[g_cols[i]]++;
        g_ptr[key[i] / (uint64_t)n + 1]++;
    }
    for (long r = 0; r < n; ++r) g_ptr[r + 1] += g_ptr[r];
}

template <int LPN, int U, int PART>
static void launch(int aux, const float* B, const int* idx, long nidx, int slices, const Parts& parts, f4* out)  // (slices by value below)
{
    const long nchunks = (nidx + CH - 1) / CH;
    unsigned grid;
    if (PART == 2) {
        grid = (unsigned)(parts.nblk * (long)slices) * 8u;  // `slices` carries the number of phases here
        slices = 1;
    } else if (PART) {
        long mx = 0;
        for (int p = 0; p < 8 / slices; ++p) mx = std::max(mx, parts.cs[p + 1] - parts.cs[p]);
        grid = (unsigned)((mx + 3) / 4) * 8u;
    } else {
        grid = (unsigned)((nchunks + 3) / 4);
        if (slices > 1) grid = (unsigned)((grid + 8 / slices - 1) / (8 / slices)) * 8u;
    }
    if (grid == 0) return;
    if (aux == 2) hipLaunchKernelGGL((k_gather<LPN, U, 2, PART>), dim3(grid), dim3(256), 0, 0, B, idx, nchunks, nidx, slices, parts, out);
    else hipLaunchKernelGGL((k_gather<LPN, U, 0, PART>), dim3(grid), dim3(256), 0, 0, B, idx, nchunks, nidx, slices, parts, out);
}

template <int PART>
static void gather(int aux, const float* B, const int* idx, long nidx, int slices, const Parts& parts, f4* out)
{
    if (nidx == 0) return;
    const int lanes = 32 / slices;
    if (lanes >= 32) launch<32, 4, PART>(aux, B, idx, nidx, slices, parts, out);
    else if (lanes == 16) launch<16, 4, PART>(aux, B, idx, nidx, slices, parts, out);
    else if (lanes == 8) launch<8, 4, PART>(aux, B, idx, nidx, slices, parts, out);
    else launch<4, 4, PART>(aux, B, idx, nidx, slices, parts, out);
}

int main(int argc, char** argv)
{
    const char* filter = argc > 1 ? argv[1] : "";
    const int scale = 20;
    const long n = 1l << scale;
    make_rmat(scale, 32);
    const long nnz = (long)g_cols.size();
    printf("R-MAT 2^%d: nnz %ld, gather volume %.2f GB per launch (512-byte rows)\n", scale, nnz, nnz * 512 / 1e9);
    std::vector<int> order(n), rank(n);
    for (long i = 0; i < n; ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return g_cnt[x] > g_cnt[y]; });
    for (long i = 0; i < n; ++i) rank[order[i]] = (int)i;
    {  // row-length census: share of the nonzeros in rows of at least T
        for (int T : {16, 32, 64, 128, 256, 1024, 4096}) {
            long rows = 0, nz = 0;
            for (long r = 0; r < n; ++r) {
                const long l = g_ptr[r + 1] - g_ptr[r];
                if (l >= T) { ++rows; nz += l; }
            }
            printf("rows with >= %4d nonzeros: %7ld (%.2f %% of the rows) hold %.1f %% of the nonzeros\n", T, rows, 100.0 * rows / n, 100.0 * nz / nnz);
        }
    }
    float* B;
    CK(hipMalloc(&B, n * 512));
    CK(hipMemset(B, 0, n * 512));
    int *d_a, *d_b;
    CK(hipMalloc(&d_a, (nnz + 8 * CH) * 4));
    CK(hipMalloc(&d_b, (nnz + 8 * CH) * 4));
    f4* out;
    CK(hipMalloc(&out, (nnz / CH + 64) * 64 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int dispatch = 0;
    auto timed = [&](const std::string& name, auto&& fn, int kernels_per_rep) -> float {
        if (*filter && name.find(filter) == std::string::npos) return 0.f;
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
        printf("%-72s min %.3f avg %.3f ms  [dispatches %d..%d, %d per rep]\n", name.c_str(), best, sum / reps, dispatch,
               dispatch + (reps + 1) * kernels_per_rep - 1, kernels_per_rep);
        dispatch += (reps + 1) * kernels_per_rep;
        fflush(stdout);
        return best;
    };
    Parts none{};
    char nm[200];
    auto tag = [&](int c, int H) { return rank[c] < H ? c : (int)((unsigned)c | 0x80000000u); };
    // reference points: the row-owned gather as shipped (S = 2, 32 768 hot slices, cold non-temporal) and S = 4 untagged
    {
        std::vector<int> h(nnz);
        for (long i = 0; i < nnz; ++i) h[i] = tag(g_cols[i], 32768);
        CK(hipMemcpy(d_a, h.data(), nnz * 4, hipMemcpyHostToDevice));
        timed("row-owned one-pass S=2 H=32768 cold nt (shipped structure)", [&] { gather<0>(2, B, d_a, nnz, 2, none, out); }, 1);
        timed("row-owned one-pass S=4 untagged", [&] { gather<0>(0, B, d_a, nnz, 4, none, out); }, 1);
    }
    for (int T : {32, 64, 128, 512}) {
        for (int P : {8, 4, 2}) {
            for (int hashed : {0, 1}) {
                if (hashed && P != 8) continue;
                // long rows: nonzeros by partition, partitions concatenated (each padded to whole chunks); short rows: as they are
                std::vector<std::vector<int>> part(P);
                std::vector<int> shortc;
                long n_long = 0;
                auto pf = [&](int c) {
                    if (hashed) return (int)(((unsigned)c * 0x9E3779B1u) >> 16) % P;
                    return rank[c] % P;
                };
                for (long r = 0; r < n; ++r) {
                    const long b = g_ptr[r], e = g_ptr[r + 1];
                    if (e - b >= T) {
                        ++n_long;
                        for (long i = b; i < e; ++i) part[pf(g_cols[i])].push_back(g_cols[i]);
                    } else {
                        for (long i = b; i < e; ++i) shortc.push_back(g_cols[i]);
                    }
                }
                long nl = 0, mn = 1l << 60, mx = 0;
                for (int p = 0; p < P; ++p) {
                    nl += (long)part[p].size();
                    mn = std::min(mn, (long)part[p].size());
                    mx = std::max(mx, (long)part[p].size());
                }
                const long ns = (long)shortc.size();
                printf("--- T=%d P=%d %s: %ld long rows, %.1f %% of the nonzeros; partitions %ld..%ld nonzeros; partials %.0f MB written + read\n",
                       T, P, hashed ? "hash" : "rank", n_long, 100.0 * nl / nnz, mn, mx, 2.0 * n_long * P * 512.0 / 1e6);
                for (int H : {0, 65536, 131072}) {
                    std::vector<int> cat;
                    Parts ps{};
                    for (int p = 0; p < P; ++p) {
                        ps.cs[p] = (long)cat.size() / CH;
                        for (int c : part[p]) cat.push_back(H ? tag(c, H) : c);
                        while (cat.size() % CH) cat.push_back(cat.back());
                    }
                    ps.cs[P] = (long)cat.size() / CH;
                    CK(hipMemcpy(d_a, cat.data(), cat.size() * 4, hipMemcpyHostToDevice));
                    const long ncat = (long)cat.size();
                    snprintf(nm, sizeof nm, "T=%d P=%d %s long pass S=%d H=%d%s", T, P, hashed ? "hash" : "rank", 8 / P, H, H ? " cold nt" : "");
                    const float tl = timed(nm, [&] { gather<1>(H ? 2 : 0, B, d_a, ncat, 8 / P, ps, out); }, 1);
                    if (H == 0) {
                        std::vector<int> sh(ns);
                        for (long i = 0; i < ns; ++i) sh[i] = tag(shortc[i], 32768);
                        CK(hipMemcpy(d_b, sh.data(), ns * 4, hipMemcpyHostToDevice));
                        snprintf(nm, sizeof nm, "T=%d P=%d %s short pass row-owned S=2 H=32768 cold nt", T, P, hashed ? "hash" : "rank");
                        const float tsn = timed(nm, [&] { gather<0>(2, B, d_b, ns, 2, none, out); }, 1);
                        snprintf(nm, sizeof nm, "T=%d P=%d %s short pass row-owned S=4 untagged", T, P, hashed ? "hash" : "rank");
                        const float tsu = timed(nm, [&] { gather<0>(0, B, d_b, ns, 4, none, out); }, 1);
                        printf("    => long %.3f + short %.3f / %.3f ms\n", tl, tsn, tsu);
                    } else {
                        printf("    => long %.3f ms\n", tl);
                    }
                }
            }
        }
    }
    // ---- LDS copy of the hottest rows of each partition (persistent workgroups) ----
    {
        const int T = 128, P = 8;
        std::vector<std::vector<int>> part(P);
        auto pf = [&](int c) { return (int)(((unsigned)c * 0x9E3779B1u) >> 16) % P; };
        for (long r = 0; r < n; ++r) {
            const long b = g_ptr[r], e = g_ptr[r + 1];
            if (e - b >= T)
                for (long i = b; i < e; ++i) part[pf(g_cols[i])].push_back(g_cols[i]);
        }
        int* d_hot;
        CK(hipMalloc(&d_hot, 8 * 256 * 4));
        for (int H : {0, 64, 128, 192, 240}) {
            for (int U : {4, 8}) {
                // the H most referenced columns of every partition (by the global count) -> slots
                std::vector<int> hot(8 * 256, 0), slot(n, -1);
                int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (long k = 0; k < n; ++k) {
                    const int c = order[k], q = pf(c);
                    if (cnt[q] < H) { slot[c] = cnt[q]; hot[q * H + cnt[q]] = c; ++cnt[q]; }
                }
                std::vector<int> cat;
                Parts ps{};
                long served = 0, all = 0;
                for (int q = 0; q < P; ++q) {
                    ps.cs[q] = (long)cat.size() / CH;
                    for (int c : part[q]) { cat.push_back(slot[c] >= 0 ? (0x40000000 | slot[c]) : c); served += slot[c] >= 0; ++all; }
                    while (cat.size() % CH) cat.push_back(cat.back());
                }
                ps.cs[P] = (long)cat.size() / CH;
                CK(hipMemcpy(d_a, cat.data(), cat.size() * 4, hipMemcpyHostToDevice));
                CK(hipMemcpy(d_hot, hot.data(), hot.size() * 4, hipMemcpyHostToDevice));
                const long ncat = (long)cat.size();
                snprintf(nm, sizeof nm, "T=128 P=8 persistent 1024-thread workgroups, LDS copy of %d rows per partition, U=%d", H, U);
                const float tl = timed(nm, [&] {
                    if (U == 4) hipLaunchKernelGGL((k_gather_lds<4>), dim3(256), dim3(1024), (size_t)H * 512, 0, B, d_a, ps, ncat, d_hot, H, out);
                    else hipLaunchKernelGGL((k_gather_lds<8>), dim3(256), dim3(1024), (size_t)H * 512, 0, B, d_a, ps, ncat, d_hot, H, out);
                }, 1);
                printf("    => long %.3f ms; %.1f %% of the long-row references served from LDS\n", tl, 100.0 * served / all);
            }
        }
    }
    // ---- more partitions than XCDs: 8 * phases partitions, XCD x takes x, x + 8, ... in time ----
    for (int T : {128, 256, 512}) {
        for (int phases : {1, 2, 4}) {
            const int P = 8 * phases;
            std::vector<std::vector<int>> part(P);
            long n_long = 0;
            auto pf = [&](int c) { return (int)(((unsigned)c * 0x9E3779B1u) >> 16) % P; };
            for (long r = 0; r < n; ++r) {
                const long b = g_ptr[r], e = g_ptr[r + 1];
                if (e - b >= T) {
                    ++n_long;
                    for (long i = b; i < e; ++i) part[pf(g_cols[i])].push_back(g_cols[i]);
                }
            }
            std::vector<int> cat;
            Parts ps{};
            long mx = 0;
            for (int p = 0; p < P; ++p) {
                ps.cs[p] = (long)cat.size() / CH;
                for (int c : part[p]) cat.push_back(c);
                while (cat.size() % CH) cat.push_back(cat.back());
                mx = std::max(mx, (long)cat.size() / CH - ps.cs[p]);
            }
            ps.cs[P] = (long)cat.size() / CH;
            ps.nblk = (mx + 3) / 4;
            CK(hipMemcpy(d_a, cat.data(), cat.size() * 4, hipMemcpyHostToDevice));
            const long ncat = (long)cat.size();
            snprintf(nm, sizeof nm, "T=%d %d partitions (%d phases per XCD) long pass S=1", T, P, phases);
            const float tl = timed(nm, [&] { launch<32, 4, 2>(0, B, d_a, ncat, phases, ps, out); }, 1);
            printf("    => long %.3f ms, %ld long rows, partial rows %.0f MB written + read\n", tl, n_long, 2.0 * n_long * P * 512.0 / 1e6);
        }
    }
    return 0;
}
