// ctx.hip — context lifetime for libfluidaudio_hip.so.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "fa_common.h"

namespace {
// every live context, so that an allocation failure in one of them can release the idle caches of the others on the same device
std::mutex g_registry_mutex;
std::vector<fa_ctx *> g_registry;

void free_caches_locked(fa_ctx *c, bool scratch_too) {   // registry mutex held, c not in a linkage call
    fa::DeviceGuard guard(c->device);
    if (c->ahc_ws) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ahc_ws); c->ahc_ws = nullptr; c->ahc_ws_bytes = 0; }
    if (scratch_too && c->scratch) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
}

// HBM pressure on `device`: everything the library holds there WITHOUT using it goes — the buffer caches of every context (the caller's own,
// its workers and helpers, every other context: a cached buffer is idle by definition) and the linkage workspaces of the contexts that are not
// inside a linkage call.  The pointers are taken out of the contexts under the locks (a context may be destroyed by its owner the moment the
// registry lock is dropped); the hipFree calls — each waits for the device — run after the locks are released.  Returns the bytes released.
size_t release_idle_device_memory(const fa_ctx *self, const int device) {
    std::vector<void *> doomed;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        for (fa_ctx *c : g_registry) {
            if (c->device != device) continue;
            {
                std::lock_guard<std::mutex> bl(c->buf_mutex);
                for (auto &b : c->buf_free) { doomed.push_back(b.first); bytes += b.second; }
                c->buf_free.clear();
                c->buf_cached_bytes = 0;
            }
            if (c != self && !c->ws_busy && c->ahc_ws) { doomed.push_back(c->ahc_ws); bytes += c->ahc_ws_bytes; c->ahc_ws = nullptr; c->ahc_ws_bytes = 0; }
        }
    }
    if (!doomed.empty()) {
        fa::DeviceGuard guard(device);
        (void)hipDeviceSynchronize();   // the last users of a cached buffer / an idle workspace may still be in flight on their streams
        for (void *q : doomed) (void)hipFree(q);
    }
    return bytes;
}

std::atomic<int32_t> g_fault[FA_FAULT_SITES];

// The switch table: values are heap copies that are never freed (a reader may hold the pointer it loaded; fa_debug_set_switch is a test hook).
constexpr int kSwCount = static_cast<int>(fa::Sw::kCount);
const char *const kSwName[kSwCount] = {
#define FA_SW_NAME(id, name) name,
    FA_SWITCHES(FA_SW_NAME, FA_SW_NAME)
#undef FA_SW_NAME
};
std::atomic<const char *> g_sw[kSwCount];
std::once_flag g_sw_once;
bool g_debug_hooks = false;   // FLUIDAUDIO_HIP_DEBUG_HOOKS=1 in the environment of the process: fa_debug_inject_fault / fa_debug_set_switch act
const char *sw_copy(const char *v) {
    if (!v) return nullptr;
    const size_t n = strlen(v) + 1;
    char *c = static_cast<char *>(malloc(n));
    if (c) memcpy(c, v, n);
    return c;
}
void sw_read_environment() {   // the library's only look at the environment
    for (int i = 0; i < kSwCount; ++i) g_sw[i].store(sw_copy(getenv(kSwName[i])), std::memory_order_release);
    const char *h = getenv("FLUIDAUDIO_HIP_DEBUG_HOOKS");
    g_debug_hooks = h && h[0] == '1';
}
bool debug_hooks() {
    std::call_once(g_sw_once, sw_read_environment);
    return g_debug_hooks;
}
}  // namespace

namespace fa {
// -DFA_POISON_WORKSPACE (make POISON=1 -> libfluidaudio_hip_poison.so): everything a call is handed — the linkage workspace, cached device buffers,
// the scratch — is filled with 0xFF first (NaN as a double, -1 as an index, a wild address as a pointer), so that code which reads bytes it
// never wrote fails in the tests instead of working on whatever an earlier call left there (round 5 found such a read of four rounds' standing
// only because a new path reused the bytes).  The GPU suite is run once over that build.
static inline void poison(fa_ctx *ctx, void *p, size_t bytes) {
#ifdef FA_POISON_WORKSPACE
    if (p && bytes) { (void)hipMemsetAsync(p, 0xFF, bytes, ctx->stream); (void)hipStreamSynchronize(ctx->stream); }
#else
    (void)ctx; (void)p; (void)bytes;
#endif
}

fa_status ws_acquire(fa_ctx *ctx, size_t bytes) {
    // The registry mutex guards the bookkeeping (who is busy, whose idle cache may be taken) — not the allocation itself: a hipMalloc of tens
    // of gigabytes takes 0.3 - 6 s, and since round 4 several host threads of one call allocate at the same time (the groups of
    // ahc_batch_uniform_groups, pooled contexts on other GPUs).  The context is marked busy first, so nobody touches its workspace while it is
    // (re)allocated outside the lock.
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        ctx->ws_busy = true;
        if (bytes > ctx->ws_cap) return set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: %zu bytes of workspace needed, the context's cap is %zu", bytes, ctx->ws_cap);
        if (ctx->ahc_ws_bytes >= bytes) { poison(ctx, ctx->ahc_ws, ctx->ahc_ws_bytes); return FA_SUCCESS; }
    }
    if (ctx->ahc_ws) { FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ahc_ws); ctx->ahc_ws = nullptr; ctx->ahc_ws_bytes = 0; }
    hipError_t e = fault_hit(FA_FAULT_WS_MALLOC) ? hipErrorOutOfMemory : hipMalloc(&ctx->ahc_ws, bytes);
    if (e != hipSuccess) {   // HBM pressure: the buffer caches of every context on this device and the idle linkage workspaces of the others go first
        (void)hipGetLastError();
        ctx->ahc_ws = nullptr;
        (void)release_idle_device_memory(ctx, ctx->device);
        e = hipMalloc(&ctx->ahc_ws, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->ahc_ws = nullptr;
        return set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: cannot allocate %zu bytes of HBM", bytes);
    }
    ctx->ahc_ws_bytes = bytes;
    poison(ctx, ctx->ahc_ws, bytes);
    return FA_SUCCESS;
}

void ws_release(fa_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    ctx->ws_busy = false;
    if (ctx->ahc_ws && ctx->ahc_ws_bytes > ctx->ws_limit) free_caches_locked(ctx, false);
}

// ---- fault injection for the tests (fa_debug_inject_fault): the next `count` passes through a site fail
bool fault_hit(const int site) {
    if (site < 0 || site >= FA_FAULT_SITES) return false;
    int32_t have = g_fault[site].load(std::memory_order_relaxed);
    while (have > 0)   // compare-exchange: two threads racing for the last armed pass cannot drive the counter negative
        if (g_fault[site].compare_exchange_weak(have, have - 1, std::memory_order_relaxed)) return true;
    return false;
}

// ---- switches (fa_common.h): the process environment, read once
const char *sw_lookup(const Sw s) {
    std::call_once(g_sw_once, sw_read_environment);
    return g_sw[static_cast<int>(s)].load(std::memory_order_acquire);
}

// ---- buffer cache of a context (see fa_ctx::buf_free)
constexpr size_t kBufCacheLimit = static_cast<size_t>(3) << 30;   // bytes a context keeps at most; ws_limit (fa_ctx_set_workspace_limit) bounds it further
void buf_cache_flush(fa_ctx *ctx) {                               // caller holds ctx->buf_mutex
    for (auto &b : ctx->buf_free) (void)hipFree(b.first);
    ctx->buf_free.clear();
    ctx->buf_cached_bytes = 0;
}
hipError_t devbuf_take(fa_ctx *ctx, size_t bytes, void **p, size_t *cap) {
    bool cached = false;
    {
        std::lock_guard<std::mutex> lock(ctx->buf_mutex);
        int best = -1;
        for (int i = 0; i < static_cast<int>(ctx->buf_free.size()); ++i) {
            const size_t c = ctx->buf_free[i].second;
            if (c >= bytes && c <= 2 * bytes + (static_cast<size_t>(1) << 20) && (best < 0 || c < ctx->buf_free[best].second)) best = i;
        }
        if (best >= 0) {
            *p = ctx->buf_free[best].first; *cap = ctx->buf_free[best].second;
            ctx->buf_cached_bytes -= *cap;
            ctx->buf_free.erase(ctx->buf_free.begin() + best);
            cached = true;
        }
    }
    if (cached) { poison(ctx, *p, *cap); return hipSuccess; }
    const size_t want = (bytes + 255) & ~static_cast<size_t>(255);
    hipError_t e = fault_hit(FA_FAULT_DEVBUF_MALLOC) ? hipErrorOutOfMemory : hipMalloc(p, want);
    if (e != hipSuccess) {   // HBM pressure: every idle byte the library holds on this device goes (the caches of ALL its contexts, idle linkage workspaces)
        (void)hipGetLastError();
        (void)release_idle_device_memory(ctx, ctx->device);
        e = hipMalloc(p, want);
    }
    *cap = want;
    if (e == hipSuccess) poison(ctx, *p, want);
    return e;
}
void devbuf_give(fa_ctx *ctx, void *p, size_t cap) {
    {
        std::lock_guard<std::mutex> lock(ctx->buf_mutex);
        const size_t limit = ctx->ws_limit < kBufCacheLimit ? ctx->ws_limit : kBufCacheLimit;   // ws_limit 0 = keep nothing between calls
        if (ctx->buf_cached_bytes + cap <= limit && ctx->buf_free.size() < 256) {
            ctx->buf_free.emplace_back(p, cap);
            ctx->buf_cached_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

fa_status ensure_scratch(fa_ctx *ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) { poison(ctx, ctx->scratch, ctx->scratch_bytes); return FA_SUCCESS; }
    if (ctx->scratch) {
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    FA_HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    poison(ctx, ctx->scratch, bytes);
    return FA_SUCCESS;
}
}  // namespace fa

extern "C" {

const char *fa_version(void) { return "fluidaudio_hip 0.1 gfx950"; }

void fa_debug_inject_fault(int32_t site, int32_t count) {
    if (!debug_hooks()) return;   // a process that did not ask for the hooks cannot be made to degrade by a stray call
    if (site >= 0 && site < FA_FAULT_SITES) g_fault[site].store(count < 0 ? 0 : count, std::memory_order_relaxed);
}

fa_status fa_debug_set_switch(const char *name, const char *value) {
    if (!name) return FA_INVALID_ARGUMENT;
    if (!debug_hooks()) return FA_RUNTIME_ERROR;
    for (int i = 0; i < kSwCount; ++i)
        if (strcmp(name, kSwName[i]) == 0) {
#ifndef FA_AB_SWITCHES
            if (fa::sw_is_ab(static_cast<fa::Sw>(i))) return FA_INVALID_ARGUMENT;   // compiled out of this build
#endif
            const char *c = sw_copy(value);
            if (value && !c) return FA_ALLOCATION_FAILURE;
            g_sw[i].store(c, std::memory_order_release);
            return FA_SUCCESS;
        }
    return FA_INVALID_ARGUMENT;
}

int32_t fa_debug_hooks_enabled(void) { return debug_hooks() ? 1 : 0; }

fa_status fa_ctx_create(int device, void *stream, fa_ctx **out) {
    if (!out) return FA_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return FA_RUNTIME_ERROR;  // no GPU: fail loudly
    if (device < 0 || device >= count) return FA_INVALID_ARGUMENT;
    fa_ctx *ctx = new (std::nothrow) fa_ctx();
    if (!ctx) return FA_ALLOCATION_FAILURE;
    ctx->device = device;
    fa::DeviceGuard guard(device);   // the caller's current device is restored on return
    if (stream) {
        ctx->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return FA_RUNTIME_ERROR; }
        ctx->owns_stream = true;
    }
    if (const char *lim = fa::sw(fa::Sw::HIP_WORKSPACE_LIMIT)) {
        char *end = nullptr;
        const unsigned long long v = strtoull(lim, &end, 10);
        if (end != lim) ctx->ws_limit = static_cast<size_t>(v);
    }
    { std::lock_guard<std::mutex> lock(g_registry_mutex); g_registry.push_back(ctx); }
    *out = ctx;
    return FA_SUCCESS;
}

void fa_ctx_destroy(fa_ctx *ctx) {
    if (!ctx) return;
    for (auto &h : ctx->helpers) { if (h) fa_ctx_destroy(h); h = nullptr; }
    for (auto &h : ctx->workers) { if (h) fa_ctx_destroy(h); h = nullptr; }
    { std::lock_guard<std::mutex> lock(g_registry_mutex); g_registry.erase(std::remove(g_registry.begin(), g_registry.end(), ctx), g_registry.end()); }
    fa::DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    { std::lock_guard<std::mutex> lock(ctx->buf_mutex); fa::buf_cache_flush(ctx); }
    if (ctx->ahc_ws) (void)hipFree(ctx->ahc_ws);
    if (ctx->poly_taps) (void)hipFree(ctx->poly_taps);
    if (ctx->poly_rows && ctx->poly_rows_free) ctx->poly_rows_free(ctx->poly_rows);
    if (ctx->ahc_graph && ctx->ahc_graph_free) ctx->ahc_graph_free(ctx->ahc_graph);
    if (ctx->ahc_uni_graph && ctx->ahc_graph_free) ctx->ahc_graph_free(ctx->ahc_uni_graph);
    if (ctx->mel_cache && ctx->mel_cache_free) ctx->mel_cache_free(ctx->mel_cache);
    for (auto &e : ctx->ahc_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->tim_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Workspace policy.  A context keeps its linkage workspace (N^2 * 8 B: 15 GB at 43 200 rows, 20 GB at 50 000) and its scratch
// buffer between calls.  fa_ctx_set_workspace_limit: a linkage workspace larger than `bytes` is released when the call that used it
// returns (0 = never keep one; default: keep, or FLUIDAUDIO_HIP_WORKSPACE_LIMIT).  fa_ctx_trim releases everything cached now.
// fa_ctx_workspace_bytes reports what is cached.  The calling thread must be the one using the context (contexts are not shared).
fa_status fa_ctx_set_workspace_limit(fa_ctx *ctx, size_t bytes) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    for (fa_ctx *h : ctx->helpers) if (h) (void)fa_ctx_set_workspace_limit(h, bytes);
    for (fa_ctx *h : ctx->workers) if (h) (void)fa_ctx_set_workspace_limit(h, bytes);
    std::vector<void *> doomed;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        ctx->ws_limit = bytes;
        if (!ctx->ws_busy && ctx->ahc_ws && ctx->ahc_ws_bytes > bytes) free_caches_locked(ctx, false);
        std::lock_guard<std::mutex> bl(ctx->buf_mutex);   // the buffer cache obeys the same limit
        while (!ctx->buf_free.empty() && ctx->buf_cached_bytes > bytes) {
            doomed.push_back(ctx->buf_free.back().first);
            ctx->buf_cached_bytes -= ctx->buf_free.back().second;
            ctx->buf_free.pop_back();
        }
    }
    if (!doomed.empty()) {
        fa::DeviceGuard guard(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (void *q : doomed) (void)hipFree(q);
    }
    return FA_SUCCESS;
}

fa_status fa_ctx_set_workspace_cap(fa_ctx *ctx, size_t bytes) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    for (fa_ctx *h : ctx->helpers) if (h) (void)fa_ctx_set_workspace_cap(h, bytes);
    for (fa_ctx *h : ctx->workers) if (h) (void)fa_ctx_set_workspace_cap(h, bytes);
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    ctx->ws_cap = bytes;
    return FA_SUCCESS;
}

fa_status fa_ctx_trim(fa_ctx *ctx) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    for (fa_ctx *h : ctx->helpers) if (h) (void)fa_ctx_trim(h);
    for (fa_ctx *h : ctx->workers) if (h) (void)fa_ctx_trim(h);
    // the registry lock guards the bookkeeping only: the pointers are taken out under it, the stream synchronisation and the hipFree calls (each
    // waits for the whole device) happen after it is dropped — another context's ws_acquire / ws_release never waits for this context's kernels
    std::vector<void *> doomed;
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        if (ctx->ws_busy) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "trim: the context is inside a linkage call");
        if (ctx->ahc_ws) { doomed.push_back(ctx->ahc_ws); ctx->ahc_ws = nullptr; ctx->ahc_ws_bytes = 0; }
        if (ctx->scratch) { doomed.push_back(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        std::lock_guard<std::mutex> bl(ctx->buf_mutex);
        for (auto &b : ctx->buf_free) doomed.push_back(b.first);
        ctx->buf_free.clear();
        ctx->buf_cached_bytes = 0;
    }
    {
        fa::DeviceGuard guard(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        for (void *q : doomed) (void)hipFree(q);
    }
    if (ctx->mel_cache && ctx->mel_cache_free) { ctx->mel_cache_free(ctx->mel_cache); ctx->mel_cache = nullptr; }
    return FA_SUCCESS;
}

size_t fa_ctx_workspace_bytes(const fa_ctx *ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    size_t total = ctx->ahc_ws_bytes + ctx->scratch_bytes + ctx->buf_cached_bytes;
    for (const fa_ctx *h : ctx->helpers) if (h) total += h->ahc_ws_bytes + h->scratch_bytes + h->buf_cached_bytes;
    for (const fa_ctx *h : ctx->workers) if (h) total += h->ahc_ws_bytes + h->scratch_bytes + h->buf_cached_bytes;
    return total;
}

// Device-time bracket of the entries that support it (today: fa_ctc_beam_search_batch_dev): events on the context's stream around the
// device work of a call, after its allocations.  fa_ctx_last_device_ms: milliseconds of the last bracketed call, < 0 if none.
fa_status fa_ctx_set_timing(fa_ctx *ctx, int32_t enable) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    fa::DeviceGuard guard(ctx->device);
    if (enable) for (auto &e : ctx->tim_ev) if (!e) FA_HIP_TRY(ctx, hipEventCreate(&e));
    ctx->timing = enable != 0;
    ctx->last_device_ms = -1.0;
    return FA_SUCCESS;
}
double fa_ctx_last_device_ms(const fa_ctx *ctx) { return ctx ? ctx->last_device_ms : -1.0; }

// The shader clock the device runs at right now: one wavefront spins for ~`spin_us` microseconds of the constant 100 MHz counter
// (s_memrealtime) and counts shader cycles (s_memtime) meanwhile.  A latency-bound kernel (the beam search walk, the merge chain) scales
// with this clock; bench.py prints it next to such legs so that a run at idle clocks can be told from a slow kernel.
namespace {
__global__ void sclk_probe_kernel(unsigned long long *out, const unsigned long long ticks) {
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) r1 = wall_clock64();
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
}  // namespace
fa_status fa_debug_sclk_mhz(fa_ctx *ctx, int32_t spin_us, double *mhz) {
    if (!ctx || !mhz) return FA_INVALID_ARGUMENT;
    *mhz = 0.0;
    if (spin_us < 1) spin_us = 1;
    if (spin_us > 100000) spin_us = 100000;
    fa::DeviceGuard guard(ctx->device);
    FA_TRY(fa::ensure_scratch(ctx, 256));
    unsigned long long *d = static_cast<unsigned long long *>(ctx->scratch);
    hipLaunchKernelGGL(sclk_probe_kernel, dim3(1), dim3(64), 0, ctx->stream, d, static_cast<unsigned long long>(spin_us) * 100ull);
    FA_HIP_TRY(ctx, hipGetLastError());
    unsigned long long h[2] = {0, 0};
    FA_HIP_TRY(ctx, hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (h[1] > 0) *mhz = 100.0 * static_cast<double>(h[0]) / static_cast<double>(h[1]);
    return FA_SUCCESS;
}

fa_status fa_ctx_synchronize(fa_ctx *ctx) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FA_SUCCESS;
}

// Pinned (page-locked) host memory: buffers allocated here are DMA targets, so the host-pointer entries can move them at the full
// PCIe rate and overlap uploads with downloads (fa_mel_batch).  Plain malloc'ed buffers keep working (staged copies).
void *fa_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void fa_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

void *fa_ctx_stream(const fa_ctx *ctx) { return ctx ? static_cast<void *>(ctx->stream) : nullptr; }

const char *fa_ctx_last_error(const fa_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

}  // extern "C"
