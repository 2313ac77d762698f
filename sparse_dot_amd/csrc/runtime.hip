// runtime.hip -- errors, per-thread context, device memory, service entry points, scan.
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <map>
#include <thread>

#include "common.hpp"

namespace mi {

// ---- errors ------------------------------------------------------------------------------------
static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }
void clear_error() { g_err[0] = 0; }

void fail(int status, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    throw status_error{status};
}

void launch_too_large(unsigned long long threads)
{
    fail(MI_SPARSE_STATUS_NOT_SUPPORTED, "internal launch of %llu threads exceeds the 2^32 per-grid limit (operand too large)", threads);
}

// ---- device memory -----------------------------------------------------------------------------
namespace {
constexpr int POOL_MAX_DEVICES = 64;
struct BlockPool {
    std::mutex m;
    std::multimap<size_t, void*> blocks[POOL_MAX_DEVICES];  // size -> cached block, per device
    size_t cached = 0;
    size_t cap = 0;  // resolved lazily from pool_max_mb / the device memory
};
BlockPool& pool()
{
    static BlockPool* p = new BlockPool();  // leaked on purpose: HIP may be gone at static destruction
    return *p;
}

// small requests: next power of two (exact-size classes); large ones: next multiple of 2 MiB
size_t pool_round(size_t n)
{
    if (n <= (size_t(1) << 20)) {
        size_t r = 256;
        while (r < n) r <<= 1;
        return r;
    }
    const size_t g = size_t(2) << 20;
    return (n + g - 1) / g * g;
}

void pool_trim_locked(BlockPool& bp)
{
    for (auto& per_dev : bp.blocks) {
        for (auto& kv : per_dev) (void)hipFree(kv.second);
        per_dev.clear();
    }
    bp.cached = 0;
}
}  // namespace

void pool_reset_cap()
{
    BlockPool& bp = pool();
    std::lock_guard<std::mutex> lk(bp.m);
    bp.cap = 0;
}

void pool_trim()
{
    BlockPool& bp = pool();
    std::lock_guard<std::mutex> lk(bp.m);
    pool_trim_locked(bp);
}

void DevBuf::alloc(size_t n)
{
    release();
    Context& c = ctx();
    c.ensure();
    const size_t want = pool_round(n ? n : 16);
    const int d = c.device;
    {
        int cur = d;
        if (hipGetDevice(&cur) == hipSuccess && cur != d)
            fail(MI_SPARSE_STATUS_INVALID_VALUE,
                 "the calling thread's current HIP device is %d but its mi_sparse context is bound to device %d; "
                 "call mi_sparse_set_device(%d) after switching devices", cur, d, cur);
    }
    BlockPool& bp = pool();
    if (d >= 0 && d < POOL_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(bp.m);
        auto it = bp.blocks[d].lower_bound(want);
        if (it != bp.blocks[d].end() && it->first <= want + want / 8) {  // best fit, at most 12.5 % slack
            p = it->second;
            bytes = it->first;
            dev = d;
            bp.cached -= it->first;
            bp.blocks[d].erase(it);
            return;
        }
    }
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {  // give the cache back to the driver and try once more
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        fail(MI_SPARSE_STATUS_ALLOC_FAILED, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    bytes = want;
    dev = d;
}

void DevBuf::release()
{
    if (!p) return;
    void* q = p;
    const size_t n = bytes;
    p = nullptr;
    bytes = 0;
    int cur = -1;
    if (options().pool_enable && dev >= 0 && dev < POOL_MAX_DEVICES && hipGetDevice(&cur) == hipSuccess && cur == dev) {
        BlockPool& bp = pool();
        {
            std::lock_guard<std::mutex> lk(bp.m);
            if (!bp.cap) {
                const int64_t mb = options().pool_max_mb;
                size_t free_b = 0, total_b = 0;
                if (mb >= 0)
                    bp.cap = (size_t)mb << 20;
                else if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                    bp.cap = total_b / 2;
                if (!bp.cap) bp.cap = 1;  // resolved: nothing is cached
            }
            if (bp.cached + n > bp.cap) goto direct;
        }
        // same guarantee hipFree gives: nothing enqueued on the device still uses the block once it can be reused
        if (hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            goto direct;
        }
        {
            std::lock_guard<std::mutex> lk(bp.m);
            bp.blocks[dev].emplace(n, q);
            bp.cached += n;
        }
        return;
    }
direct:
    (void)hipFree(q);
}

// ---- asynchronous device -> host words ------------------------------------------------------------
namespace {
// page-locked 64-byte slots carved out of one hipHostMalloc'd slab per 1024 slots: pinning host memory costs
// ~1 ms per call, far too much to pay per plan
struct PinnedSlots {
    std::mutex m;
    std::vector<int64_t*> free_list;
    int64_t* take()
    {
        std::lock_guard<std::mutex> lk(m);
        if (free_list.empty()) {
            void* slab = nullptr;
            MI_HIP_CHECK(hipHostMalloc(&slab, 1024 * 64, hipHostMallocDefault));
            for (int i = 0; i < 1024; ++i) free_list.push_back(reinterpret_cast<int64_t*>(static_cast<char*>(slab) + 64 * i));
        }
        int64_t* p = free_list.back();
        free_list.pop_back();
        return p;
    }
    void give(int64_t* p)
    {
        std::lock_guard<std::mutex> lk(m);
        free_list.push_back(p);
    }
};
PinnedSlots& pinned_slots()
{
    static PinnedSlots* s = new PinnedSlots();  // leaked on purpose (HIP may be gone at static destruction)
    return *s;
}
}  // namespace

AsyncWord::~AsyncWord()
{
    if (ev) {
        if (pending) (void)hipEventSynchronize(ev);  // the copy must not land in a slot that has been handed on
        (void)hipEventDestroy(ev);
    }
    if (host) pinned_slots().give(host);
}

void AsyncWord::ensure()
{
    if (host) return;
    host = pinned_slots().take();
    memset(host, 0, 8 * sizeof(int64_t));
    MI_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
}

void AsyncWord::post(const void* dev_src, size_t bytes, hipStream_t s)
{
    ensure();
    MI_HIP_CHECK(hipMemcpyAsync(host, dev_src, bytes, hipMemcpyDeviceToHost, s));
    MI_HIP_CHECK(hipEventRecord(ev, s));
    pending = true;
}

bool AsyncWord::ready()
{
    if (!pending) return false;
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) {
        pending = false;
        return true;
    }
    if (e != hipErrorNotReady) (void)hipGetLastError();
    return false;
}

// ---- context -----------------------------------------------------------------------------------
static thread_local Context* g_ctx = nullptr;

Context& ctx()
{
    if (!g_ctx) g_ctx = new Context();  // intentionally leaked at thread exit (HIP may be gone)
    return *g_ctx;
}

void Context::ensure()
{
    if (initialised) return;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fail(MI_SPARSE_STATUS_EXECUTION_FAILED,
             "no HIP device available (%s); libmi_sparse has no CPU path",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0) {
        // no mi_sparse_set_device on this thread: work on whatever device the caller made current (torch.cuda.set_device,
        // hipSetDevice, ROCR_VISIBLE_DEVICES ...) instead of silently moving the thread to device 0
        int cur = 0;
        MI_HIP_CHECK(hipGetDevice(&cur));
        device = cur;
    }
    if (device >= n) fail(MI_SPARSE_STATUS_INVALID_VALUE, "device %d out of range (%d devices)", device, n);
    MI_HIP_CHECK(hipSetDevice(device));
    int n_cu = 0;
    MI_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
    cus = n_cu > 0 ? n_cu : 256;
    size_t free_b = 0, total_b = 0;
    MI_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
    total_bytes = total_b;
    initialised = true;
}

size_t device_total_bytes()
{
    Context& c = ctx();
    c.ensure();
    return c.total_bytes;
}

void* Context::scratch_alloc(size_t bytes)
{
    ensure();
    const size_t off = (scratch_used + 255) & ~size_t(255);
    const size_t need = off + (bytes ? bytes : 1);
    if (need > scratch.bytes) {
        // grow: kernels already enqueued (by this call or an earlier one) may still read the old
        // arena, so it is retired -- kept alive until the next synchronise -- and a new one started
        size_t cap = scratch.bytes ? scratch.bytes : (size_t(1) << 20);
        while (cap < need) cap *= 2;
        retired.push_back(std::move(scratch));
        scratch.alloc(cap);
        scratch_used = (bytes ? bytes : 1);
        return scratch.p;
    }
    scratch_used = need;
    return static_cast<char*>(scratch.p) + off;
}

// make room for `bytes` more in ONE step and at their exact size (a caller that knows its total: the doubling above would
// retire 1 + 2 + 4 + ... GB arenas on the way to 8 -- and keep them until the next synchronise)
void Context::scratch_reserve(size_t bytes)
{
    ensure();
    const size_t off = (scratch_used + 255) & ~size_t(255);
    const size_t need = off + bytes + 4096;
    if (need <= scratch.bytes) return;
    retired.push_back(std::move(scratch));
    scratch.alloc(need);
    scratch_used = 0;
}

// give the arena back (option pool_trim): the next call starts a new one
void Context::scratch_release()
{
    if (!initialised) return;
    sync();
    scratch.release();
    scratch_used = 0;
}

void Context::sync()
{
    ensure();
    MI_HIP_CHECK(hipStreamSynchronize(stream));
    retired.clear();
}

// ---- parallel staged copies of pageable host memory ------------------------------------------------
namespace {
class CopyEngine {
public:
    static constexpr size_t CHUNK = size_t(4) << 20;
    static CopyEngine& get()
    {
        static CopyEngine* e = new CopyEngine();  // leaked on purpose (worker threads outlive static destruction)
        return *e;
    }
    // kind 0: host -> device, 1: device -> host.  Blocks until every chunk has been handed to / taken from the DMA.
    void run(int kind, char* host, char* dev, size_t n, int device, hipStream_t user_stream)
    {
        std::lock_guard<std::mutex> serial(api_);  // one transfer at a time: the slots belong to it
        start_workers();
        hipEvent_t after = nullptr;
        if (kind == 1) {  // the data is produced by work enqueued on the caller's stream
            MI_HIP_CHECK(hipEventCreateWithFlags(&after, hipEventDisableTiming));
            MI_HIP_CHECK(hipEventRecord(after, user_stream));
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = Job{kind, host, dev, n, device, after};
            next_chunk_.store(0);
            nchunks_ = (n + CHUNK - 1) / CHUNK;
            pending_ = (int)workers_.size();
            failed_ = false;
            ++generation_;
        }
        cv_.notify_all();
        {
            std::unique_lock<std::mutex> lk(m_);
            done_cv_.wait(lk, [&] { return pending_ == 0; });
        }
        if (after) (void)hipEventDestroy(after);
        if (failed_) fail(MI_SPARSE_STATUS_EXECUTION_FAILED, "staged host/device copy failed: %s", err_);
        if (kind == 0)  // order the caller's stream behind every worker's last transfer
            for (auto& w : workers_)
                if (w->last_valid) MI_HIP_CHECK(hipStreamWaitEvent(user_stream, w->last, 0));
    }

private:
    struct Job {
        int kind;
        char* host;
        char* dev;
        size_t n;
        int device;
        hipEvent_t after;
    };
    struct Worker {
        std::thread th;
        int device = -1;
        hipStream_t s = nullptr;
        void* slot[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        hipEvent_t last = nullptr;
        bool last_valid = false;
        bool busy[2] = {false, false};  // an H2D transfer out of the slot has been enqueued and not yet waited for
    };
    std::mutex api_, m_;
    std::condition_variable cv_, done_cv_;
    std::vector<Worker*> workers_;
    Job job_{};
    std::atomic<size_t> next_chunk_{0};
    size_t nchunks_ = 0;
    int pending_ = 0;
    unsigned long long generation_ = 0;
    bool failed_ = false;
    char err_[160] = {0};

    void start_workers()
    {
        if (!workers_.empty()) return;
        unsigned hw = std::thread::hardware_concurrency();
        int nw = (int)(hw / 4);
        if (nw < 2) nw = 2;
        if (nw > 8) nw = 8;
        for (int i = 0; i < nw; ++i) {
            Worker* w = new Worker();
            workers_.push_back(w);
            w->th = std::thread([this, w] { loop(w); });
            w->th.detach();
        }
    }
    bool prepare(Worker* w, int device)
    {
        if (hipSetDevice(device) != hipSuccess) return false;
        if (w->device == device) return true;
        if (w->s) {  // moved to another device: rebuild the stream / events (slots are host memory, reusable)
            (void)hipStreamSynchronize(w->s);
            w->busy[0] = w->busy[1] = false;
            (void)hipStreamDestroy(w->s);
            for (int k = 0; k < 2; ++k) (void)hipEventDestroy(w->ev[k]);
            (void)hipEventDestroy(w->last);
        }
        if (hipStreamCreateWithFlags(&w->s, hipStreamNonBlocking) != hipSuccess) return false;
        for (int k = 0; k < 2; ++k) {
            if (!w->slot[k] && hipHostMalloc(&w->slot[k], CHUNK, hipHostMallocDefault) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&w->ev[k], hipEventDisableTiming) != hipSuccess) return false;
        }
        if (hipEventCreateWithFlags(&w->last, hipEventDisableTiming) != hipSuccess) return false;
        w->device = device;
        return true;
    }
    bool work(Worker* w, const Job& j)
    {
        if (!prepare(w, j.device)) return false;
        w->last_valid = false;
        if (j.kind == 1 && hipStreamWaitEvent(w->s, j.after, 0) != hipSuccess) return false;
        size_t prev_off = 0, prev_len = 0;
        int prev_k = -1, i = 0;
        for (;;) {
            const size_t c = next_chunk_.fetch_add(1);
            if (c >= nchunks_) break;
            const size_t off = c * CHUNK, len = (off + CHUNK <= j.n) ? CHUNK : j.n - off;
            const int k = i++ & 1;
            // the slot may still feed a transfer of THIS or of an EARLIER job (H2D jobs return once enqueued)
            if (w->busy[k]) {
                if (hipEventSynchronize(w->ev[k]) != hipSuccess) return false;
                w->busy[k] = false;
            }
            if (j.kind == 0) {
                memcpy(w->slot[k], j.host + off, len);
                if (hipMemcpyAsync(j.dev + off, w->slot[k], len, hipMemcpyHostToDevice, w->s) != hipSuccess) return false;
                if (hipEventRecord(w->ev[k], w->s) != hipSuccess) return false;
                w->busy[k] = true;
            } else {
                if (hipMemcpyAsync(w->slot[k], j.dev + off, len, hipMemcpyDeviceToHost, w->s) != hipSuccess) return false;
                if (hipEventRecord(w->ev[k], w->s) != hipSuccess) return false;
                if (prev_k >= 0) {  // drain the previous chunk while this one is in flight
                    if (hipEventSynchronize(w->ev[prev_k]) != hipSuccess) return false;
                    memcpy(j.host + prev_off, w->slot[prev_k], prev_len);
                }
                prev_k = k;
                prev_off = off;
                prev_len = len;
            }
        }
        if (j.kind == 0) {
            if (i) {
                if (hipEventRecord(w->last, w->s) != hipSuccess) return false;
                w->last_valid = true;
            }
        } else if (prev_k >= 0) {
            if (hipEventSynchronize(w->ev[prev_k]) != hipSuccess) return false;
            memcpy(j.host + prev_off, w->slot[prev_k], prev_len);
        }
        return true;
    }
    void loop(Worker* w)
    {
        unsigned long long seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                j = job_;
            }
            const bool ok = work(w, j);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (!ok) {
                    failed_ = true;
                    snprintf(err_, sizeof(err_), "%s", hipGetErrorString(hipGetLastError()));
                }
                if (--pending_ == 0) done_cv_.notify_all();
            }
        }
    }
};
constexpr size_t STAGED_COPY_MIN = size_t(8) << 20;  // below this a plain (driver-staged) copy is as fast
}  // namespace

void copy_h2d(void* dst_dev, const void* src_host, size_t n)
{
    if (!n) return;
    Context& c = ctx();
    c.ensure();
    if (n >= STAGED_COPY_MIN && options().staged_copies) {
        CopyEngine::get().run(0, const_cast<char*>(static_cast<const char*>(src_host)), static_cast<char*>(dst_dev), n,
                              c.device, c.stream);
        return;
    }
    MI_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, n, hipMemcpyHostToDevice, c.stream));
}

void copy_d2h(void* dst_host, const void* src_dev, size_t n)
{
    Context& c = ctx();
    c.ensure();
    if (!n) {
        c.sync();
        return;
    }
    if (n >= STAGED_COPY_MIN && options().staged_copies) {
        CopyEngine::get().run(1, static_cast<char*>(dst_host), const_cast<char*>(static_cast<const char*>(src_dev)), n,
                              c.device, c.stream);
        c.sync();  // same post-condition as the plain path: the caller's stream is idle, retired arenas can go
        return;
    }
    MI_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, n, hipMemcpyDeviceToHost, c.stream));
    c.sync();
}

// ---- pointer location --------------------------------------------------------------------------
Loc locate(const void* p)
{
    if (!p) return Loc::Host;
    ctx().ensure();
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'd host memory is "invalid value" to HIP
        return Loc::Host;
    }
    switch (attr.type) {
        case hipMemoryTypeDevice:
        case hipMemoryTypeManaged:
        case hipMemoryTypeArray:
            return Loc::Device;
        default:
            return Loc::Host;
    }
}

void Staged::stage_in(const void* p, size_t n, bool copy_contents)
{
    bytes = n;
    if (locate(p) == Loc::Device) {
        dev = const_cast<void*>(p);
        host = nullptr;
        return;
    }
    host = const_cast<void*>(p);
    own.alloc(n);
    dev = own.p;
    if (copy_contents && n) copy_h2d(dev, p, n);
}

void Staged::copy_back()
{
    if (!host) return;
    copy_d2h(host, dev, bytes);
}

Options& options()
{
    static Options o;
    return o;
}

Counters& counters()
{
    static thread_local Counters c;
    return c;
}

static thread_local char t_last_kernel[192] = "";
void note_kernel(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_last_kernel, sizeof t_last_kernel, fmt, ap);
    va_end(ap);
}
const char* last_kernel_name() { return t_last_kernel; }

// ---- exclusive scan ----------------------------------------------------------------------------
// Three-kernel scan: per-block sums -> single-block scan of the sums -> add back.  n is at most a
// few tens of millions (row counts), so this is never the bottleneck.
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;  // per thread -> 2048 per block

__global__ void __launch_bounds__(SCAN_BLOCK) scan_block_sums(const int64_t* in, int64_t n, int64_t* sums)
{
    __shared__ int64_t red[SCAN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS;
    int64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + (int64_t)k * SCAN_BLOCK + threadIdx.x;
        if (i < n) s += in[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = SCAN_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// one block: exclusive scan of `sums` (nb entries) in place; total written to sums[nb]
__global__ void __launch_bounds__(SCAN_BLOCK) scan_sums(int64_t* sums, int64_t nb)
{
    __shared__ int64_t tile[SCAN_BLOCK];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += SCAN_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = (i < nb) ? sums[i] : 0;
        tile[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < SCAN_BLOCK; off <<= 1) {  // Hillis-Steele inclusive scan
            int64_t t = 0;
            if ((int)threadIdx.x >= off) t = tile[threadIdx.x - off];
            __syncthreads();
            tile[threadIdx.x] += t;
            __syncthreads();
        }
        const int64_t incl = tile[threadIdx.x];
        const int64_t c = carry;
        if (i < nb) sums[i] = c + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_BLOCK - 1) carry = c + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply(const int64_t* in, int64_t n, const int64_t* sums,
                                                         int64_t* out)
{
    // each thread owns SCAN_ITEMS CONSECUTIVE elements so the block scan is over thread totals
    __shared__ int64_t tile[SCAN_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t s = 0;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0;
        s += v[k];
    }
    tile[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        int64_t t = 0;
        if ((int)threadIdx.x >= off) t = tile[threadIdx.x - off];
        __syncthreads();
        tile[threadIdx.x] += t;
        __syncthreads();
    }
    int64_t run = sums[blockIdx.x] + tile[threadIdx.x] - s;
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_BLOCK - 1) out[n] = sums[gridDim.x];
}

__global__ void scan_empty(int64_t* out) { out[0] = 0; }

int64_t exclusive_scan_i64(const int64_t* in, int64_t* out, int64_t n)
{
    Context& c = ctx();
    if (n <= 0) {
        MI_LAUNCH(scan_empty, dim3(1), dim3(1), c.stream, out);
        return 0;
    }
    const int64_t nb = ceil_div(n, (int64_t)SCAN_BLOCK * SCAN_ITEMS);
    int64_t* sums = static_cast<int64_t*>(c.scratch_alloc(sizeof(int64_t) * (size_t)(nb + 1)));
    MI_LAUNCH(scan_block_sums, dim3((unsigned)nb), dim3(SCAN_BLOCK), c.stream, in, n, sums);
    MI_LAUNCH(scan_sums, dim3(1), dim3(SCAN_BLOCK), c.stream, sums, nb);
    MI_LAUNCH(scan_apply, dim3((unsigned)nb), dim3(SCAN_BLOCK), c.stream, in, n, (const int64_t*)sums, out);
    int64_t total = 0;
    MI_HIP_CHECK(hipMemcpyAsync(&total, out + n, sizeof(int64_t), hipMemcpyDeviceToHost, c.stream));
    MI_HIP_CHECK(hipStreamSynchronize(c.stream));
    return total;
}

}  // namespace mi

// ================================================================================================
// service entry points
// ================================================================================================
// The device's copy rate as THIS library can reach it: the "measured HBM roofline" the fractions in bench.py are quoted
// against.  Every lane moves 16 bytes per access (the guide's float4 copy: 6.29 TB/s of the 8 TB/s spec), UNROLL independent
// accesses in flight per lane, non-temporal both ways, a grid of a few workgroups per CU striding over the buffer.
namespace mi {
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_probe_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace mi


extern "C" {

mi_sparse_status_t mi_sparse_get_version_string(char* buf, int len)
{
    return mi::guarded([&] {
        if (!buf || len <= 0) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "bad buffer");
        int n = 0;
        char dev[300] = "no HIP device visible";
        if (hipGetDeviceCount(&n) == hipSuccess && n > 0) {
            hipDeviceProp_t prop;
            int d = mi::ctx().device;
            if (d < 0 && hipGetDevice(&d) != hipSuccess) d = 0;
            if (hipGetDeviceProperties(&prop, d) == hipSuccess)
                snprintf(dev, sizeof(dev), "%d device(s); device %d: %s (%s, %d CUs, %.0f GiB)", n, d, prop.name,
                         prop.gcnArchName, prop.multiProcessorCount,
                         (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
        } else {
            (void)hipGetLastError();
        }
        snprintf(buf, (size_t)len, "mi_sparse 0.1.0 (HIP, gfx950 / CDNA4 kernels): %s", dev);
    });
}

int mi_sparse_get_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

mi_sparse_status_t mi_sparse_set_device(int device)
{
    return mi::guarded([&] {
        mi::Context& c = mi::ctx();
        if (device < 0) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "negative device");
        if (c.initialised && c.device != device) {
            c.sync();
            c.scratch.release();
            c.scratch_used = 0;
            c.initialised = false;
        }
        c.device = device;
        c.ensure();
    });
}

mi_sparse_status_t mi_sparse_set_stream(void* hip_stream)
{
    return mi::guarded([&] {
        mi::Context& c = mi::ctx();
        hipStream_t s = static_cast<hipStream_t>(hip_stream);
        if (c.initialised && s != c.stream) c.sync();  // the scratch arena is ordered by ONE stream
        c.stream = s;
    });
}

int mi_sparse_get_device(void) { return mi::ctx().initialised ? mi::ctx().device : -1; }

mi_sparse_status_t mi_sparse_get_stream(void** hip_stream)
{
    return mi::guarded([&] {
        if (!hip_stream) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL stream pointer");
        *hip_stream = static_cast<void*>(mi::ctx().stream);
    });
}

mi_sparse_status_t mi_sparse_synchronize(void)
{
    return mi::guarded([&] { mi::ctx().sync(); });
}

const char* mi_sparse_last_error(void) { return mi::get_error(); }

mi_sparse_status_t mi_sparse_set_option(const char* name, int64_t value)
{
    return mi::guarded([&] {
        if (!name) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "NULL option name");
        mi::Options& o = mi::options();
        if (!strcmp(name, "spmm_chunk")) {
            if (value != 128 && value != 256 && value != 512 && value != 1024)
                mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_chunk must be 128, 256, 512 or 1024");
            o.spmm_chunk = value;
        } else if (!strcmp(name, "spmm_unroll")) {
            if (value != 4 && value != 8) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_unroll must be 4 or 8");
            o.spmm_unroll = value;
        } else if (!strcmp(name, "spmm_hot_kb")) {
            if (value < 0) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_hot_kb must be >= 0");
            o.spmm_hot_kb = value;
        } else if (!strcmp(name, "gemm_big_tiles")) {
            o.gemm_big_tiles = value;
        } else if (!strcmp(name, "spmm_kpart")) {
            if (value < 0 || value > 2) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_kpart must be 0, 1 or 2");
            o.spmm_kpart = value;
        } else if (!strcmp(name, "spmm_kpart_min_row")) {
            if (value < 2) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_kpart_min_row must be >= 2");
            o.spmm_kpart_min_row = value;
        } else if (!strcmp(name, "spmm_kpart_chunk")) {
            if (value != 128 && value != 256 && value != 512 && value != 1024)
                mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_kpart_chunk must be 128, 256, 512 or 1024");
            o.spmm_kpart_chunk = value;
        } else if (!strcmp(name, "spmm_kpart_parts")) {
            if (value != 8 && value != 4 && value != 2) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_kpart_parts must be 8, 4 or 2");
            o.spmm_kpart_parts = value;
        } else if (!strcmp(name, "spmm_slices")) {
            if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
                mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spmm_slices must be 0 (automatic), 1, 2, 4 or 8");
            o.spmm_slices = value;
        } else if (!strcmp(name, "gram_heads")) {
            o.gram_heads = value;
        } else if (!strcmp(name, "gram_sliced")) {
            o.gram_sliced = value;
        } else if (!strcmp(name, "gram_persistent")) {
            o.gram_persistent = value;
        } else if (!strcmp(name, "gram_tile_kb")) {
            if (value != 0 && value != 64 && value != 128 && value != 152)
                mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "gram_tile_kb must be 0 (automatic), 64, 128 or 152");
            o.gram_tile_kb = value;
        } else if (!strcmp(name, "bsr_native")) {
            o.bsr_native = value;
        } else if (!strcmp(name, "staged_copies")) {
            o.staged_copies = value;
        } else if (!strcmp(name, "spmm_plan_sync")) {
            o.spmm_plan_sync = value;
        } else if (!strcmp(name, "spmm_hot_force")) {
            o.spmm_hot_force = value;
        } else if (!strcmp(name, "spmm_force_generic")) {
            o.spmm_force_generic = value;
        } else if (!strcmp(name, "spgemm_force_global")) {
            o.spgemm_force_global = value;
        } else if (!strcmp(name, "spgemm_part_log2s_bias")) {
            o.spgemm_part_log2s_bias = value;
        } else if (!strcmp(name, "spgemm_slice_table")) {
            o.spgemm_slice_table = value;
        } else if (!strcmp(name, "spgemm_slice_table_max")) {
            o.spgemm_slice_table_max = value;
        } else if (!strcmp(name, "spgemm_lds_parts")) {
            o.spgemm_lds_parts = value;
        } else if (!strcmp(name, "spgemm_global_mode")) {
            o.spgemm_global_mode = value;
        } else if (!strcmp(name, "spgemm_group")) {
            o.spgemm_group = value;
        } else if (!strcmp(name, "spgemm_rank")) {
            o.spgemm_rank = value;
        } else if (!strcmp(name, "sort_ranges")) {
            o.sort_ranges = value;
        } else if (!strcmp(name, "transpose_radix")) {
            o.transpose_radix = value;
        } else if (!strcmp(name, "transpose_radix_bits")) {
            o.transpose_radix_bits = value;
        } else if (!strcmp(name, "transpose_lds_hist")) {
            o.transpose_lds_hist = value;
        } else if (!strcmp(name, "spgemm_narrow_ptr")) {
            o.spgemm_narrow_ptr = value;
        } else if (!strcmp(name, "spmmd_lds")) {
            o.spmmd_lds = value;
        } else if (!strcmp(name, "spgemm_col_panels")) {
            o.spgemm_col_panels = value;
        } else if (!strcmp(name, "spgemm_sort_ingest")) {
            o.spgemm_sort_ingest = value;
        } else if (!strcmp(name, "spgemm_onepass")) {
            o.spgemm_onepass = value;
        } else if (!strcmp(name, "spgemm_hub")) {
            if (value < 0 || value > 3) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spgemm_hub must be 0, 1, 2 or 3");
            o.spgemm_hub = value;
        } else if (!strcmp(name, "spgemm_hub_min_products")) {
            o.spgemm_hub_min_products = value;
        } else if (!strcmp(name, "spgemm_hub_fill_pct")) {
            if (value < 1 || value > 1000) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spgemm_hub_fill_pct must be in [1, 1000]");
            o.spgemm_hub_fill_pct = value;
        } else if (!strcmp(name, "spgemm_hub_acc_kb")) {
            if (value < 1 || value > 128) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spgemm_hub_acc_kb must be in [1, 128]");
            o.spgemm_hub_acc_kb = value;
        } else if (!strcmp(name, "spgemm_hub_block_kb")) {
            if (value < 1) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "spgemm_hub_block_kb must be >= 1");
            o.spgemm_hub_block_kb = value;
        } else if (!strcmp(name, "pool_enable")) {
            o.pool_enable = value;
            if (!value) mi::pool_trim();
        } else if (!strcmp(name, "pool_max_mb")) {
            o.pool_max_mb = value;
            mi::pool_trim();
            mi::pool_reset_cap();
        } else if (!strcmp(name, "pool_trim")) {
            mi::ctx().scratch_release();  // this thread's scratch arena too (it only ever grows otherwise)
            mi::pool_trim();
        } else if (!strcmp(name, "trace_phases")) {
            o.trace_phases = value;
        } else if (!strcmp(name, "profile_events")) {
            o.profile_events = value;
        } else if (!strcmp(name, "gram_queue")) {
            o.gram_queue = value;
        } else if (!strcmp(name, "deterministic")) {
            o.deterministic = value ? 1 : 0;
        } else {
            mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "unknown option '%s'", name);
        }
    });
}

mi_sparse_status_t mi_sparse_get_last_kernel(char* buf, int len)
{
    return mi::guarded([&] {
        if (!buf || len <= 0) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL buffer");
        snprintf(buf, (size_t)len, "%s", mi::last_kernel_name());
    });
}

mi_sparse_status_t mi_sparse_probe_copy(int64_t bytes, int reps, double* best_gbs)
{
    return mi::guarded([&] {
        if (!best_gbs || bytes < (1 << 20) || reps < 1) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "probe_copy: bytes >= 1 MiB, reps >= 1, non-NULL result");
        mi::Context& c = mi::ctx();
        c.ensure();
        mi::DevBuf a, b;
        const size_t n16 = (size_t)bytes / 16;
        a.alloc(n16 * 16);
        b.alloc(n16 * 16);
        MI_HIP_CHECK(hipMemsetAsync(a.p, 1, n16 * 16, c.stream));
        MI_HIP_CHECK(hipMemsetAsync(b.p, 2, n16 * 16, c.stream));
        hipEvent_t e0, e1;
        MI_HIP_CHECK(hipEventCreate(&e0));
        MI_HIP_CHECK(hipEventCreate(&e1));
        double best = 0.0;
        // variants: one workgroup per 256 x unroll accesses ("cover": no loop; measured best -- 5.9 - 6.0 TB/s non-temporal, unroll 4,
        // against 4.4 - 4.8 for persistent grids of 4 - 16 workgroups per CU, profiles/r06_copy_probe.log) with unroll 1 / 2 / 4 / 8,
        // plain and non-temporal, and two persistent grids for the record.  MI_PROBE_TRACE=1 prints every variant.
        const bool trace = getenv("MI_PROBE_TRACE") != nullptr;
        struct Variant { int wpc, unroll; bool nt; };
        const Variant variants[] = {{0, 1, false}, {0, 1, true}, {0, 2, false}, {0, 2, true}, {0, 4, false}, {0, 4, true},
                                    {0, 8, false}, {0, 8, true}, {8, 4, true}, {16, 4, true}};
        for (const Variant& vr : variants) {
            const size_t cover = (n16 + (size_t)256 * vr.unroll - 1) / ((size_t)256 * vr.unroll);
            const unsigned grid = vr.wpc ? (unsigned)(c.cus * vr.wpc) : (unsigned)std::min<size_t>(cover, (size_t)1 << 30);
            const mi::u32x4* src = (const mi::u32x4*)a.p;
            mi::u32x4* dst = (mi::u32x4*)b.p;
            for (int r = 0; r < reps + 1; ++r) {
                MI_HIP_CHECK(hipEventRecord(e0, c.stream));
#define MI_PROBE_GO(U)                                                                       \
    if (vr.nt) mi::k_probe_copy<U, true><<<grid, 256, 0, c.stream>>>(src, dst, n16);        \
    else mi::k_probe_copy<U, false><<<grid, 256, 0, c.stream>>>(src, dst, n16)
                if (vr.unroll == 1) { MI_PROBE_GO(1); }
                else if (vr.unroll == 2) { MI_PROBE_GO(2); }
                else if (vr.unroll == 4) { MI_PROBE_GO(4); }
                else { MI_PROBE_GO(8); }
#undef MI_PROBE_GO
                MI_HIP_CHECK(hipEventRecord(e1, c.stream));
                MI_HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                MI_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r && ms > 0.f) best = std::max(best, 2.0 * (double)(n16 * 16) / ((double)ms * 1e6));
                if (r && trace)
                    fprintf(stderr, "[mi_sparse probe_copy] %s wg/cu, unroll %d, %s: %.0f GB/s\n", vr.wpc ? std::to_string(vr.wpc).c_str() : "cover",
                            vr.unroll, vr.nt ? "non-temporal" : "plain", 2.0 * (double)(n16 * 16) / ((double)ms * 1e6));
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        MI_HIP_CHECK(hipGetLastError());
        *best_gbs = best;
    });
}

mi_sparse_status_t mi_sparse_get_counter(const char* name, double* value)
{
    return mi::guarded([&] {
        if (!name) mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "NULL counter name");
        mi::Counters& k = mi::counters();
        if (!strcmp(name, "reset")) {
            k = mi::Counters();
            if (value) *value = 0.0;
            return;
        }
        if (!value) mi::fail(MI_SPARSE_STATUS_NOT_INITIALIZED, "NULL value pointer");
        if (!strcmp(name, "spmm_kernel_ms")) *value = k.spmm_kernel_ms;
        else if (!strcmp(name, "spmm_kernel_launches")) *value = k.spmm_kernel_launches;
        else if (!strcmp(name, "spmm_last_tagged")) *value = k.spmm_last_tagged;
        else if (!strcmp(name, "spmm_hot_coverage")) *value = k.spmm_hot_coverage;
        else if (!strcmp(name, "spmm_last_slices")) *value = k.spmm_last_slices;
        else if (!strcmp(name, "spmm_plan_ms")) *value = k.spmm_plan_ms;
        else if (!strcmp(name, "spmm_plans_built")) *value = k.spmm_plans_built;
        else if (!strcmp(name, "spmm_last_kpart")) *value = k.spmm_last_kpart;
        else if (!strcmp(name, "spmm_kpart_long_share")) *value = k.spmm_kpart_long_share;
        else if (!strcmp(name, "spmm_kpart_build_ms")) *value = k.spmm_kpart_build_ms;
        else if (!strcmp(name, "spgemm_panels")) *value = k.spgemm_panels;
        else if (!strcmp(name, "spgemm_hub_items")) *value = k.spgemm_hub_items;
        else if (!strcmp(name, "bsr_native_calls")) *value = k.bsr_native_calls;
        else mi::fail(MI_SPARSE_STATUS_INVALID_VALUE, "unknown counter '%s'", name);
    });
}

}  // extern "C"
