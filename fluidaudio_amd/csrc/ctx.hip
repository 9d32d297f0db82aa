// ctx.hip — context lifetime for libfluidaudio_hip.so.
#include <algorithm>
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
}  // namespace

namespace fa {
fa_status ws_acquire(fa_ctx *ctx, size_t bytes) {
    // The registry mutex guards the bookkeeping (who is busy, whose idle cache may be taken) — not the allocation itself: a hipMalloc of tens
    // of gigabytes takes 0.3 - 6 s, and since round 4 several host threads of one call allocate at the same time (the groups of
    // ahc_batch_uniform_groups, pooled contexts on other GPUs).  The context is marked busy first, so nobody touches its workspace while it is
    // (re)allocated outside the lock.
    {
        std::lock_guard<std::mutex> lock(g_registry_mutex);
        ctx->ws_busy = true;
        if (bytes > ctx->ws_cap) return set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: %zu bytes of workspace needed, the context's cap is %zu", bytes, ctx->ws_cap);
        if (ctx->ahc_ws_bytes >= bytes) return FA_SUCCESS;
    }
    if (ctx->ahc_ws) { FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ahc_ws); ctx->ahc_ws = nullptr; ctx->ahc_ws_bytes = 0; }
    hipError_t e = hipMalloc(&ctx->ahc_ws, bytes);
    if (e != hipSuccess) {   // HBM pressure: the idle linkage workspaces of the other contexts on this device go first (their scratch stays)
        (void)hipGetLastError();
        ctx->ahc_ws = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_registry_mutex);
            for (fa_ctx *c : g_registry)
                if (c != ctx && c->device == ctx->device && !c->ws_busy && c->ahc_ws) free_caches_locked(c, false);
        }
        e = hipMalloc(&ctx->ahc_ws, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->ahc_ws = nullptr;
        return set_error(ctx, FA_ALLOCATION_FAILURE, "ahc: cannot allocate %zu bytes of HBM", bytes);
    }
    ctx->ahc_ws_bytes = bytes;
    return FA_SUCCESS;
}

void ws_release(fa_ctx *ctx) {
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    ctx->ws_busy = false;
    if (ctx->ahc_ws && ctx->ahc_ws_bytes > ctx->ws_limit) free_caches_locked(ctx, false);
}

// ---- buffer cache of a context (see fa_ctx::buf_free)
constexpr size_t kBufCacheLimit = static_cast<size_t>(3) << 30;   // bytes a context keeps; beyond that a returned buffer is released
void buf_cache_flush(fa_ctx *ctx) {                               // caller holds ctx->buf_mutex
    for (auto &b : ctx->buf_free) (void)hipFree(b.first);
    ctx->buf_free.clear();
    ctx->buf_cached_bytes = 0;
}
hipError_t devbuf_take(fa_ctx *ctx, size_t bytes, void **p, size_t *cap) {
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
            return hipSuccess;
        }
    }
    const size_t want = (bytes + 255) & ~static_cast<size_t>(255);
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {   // HBM pressure: this context's own cache goes first
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lock(ctx->buf_mutex); buf_cache_flush(ctx); }
        e = hipMalloc(p, want);
    }
    *cap = want;
    return e;
}
void devbuf_give(fa_ctx *ctx, void *p, size_t cap) {
    {
        std::lock_guard<std::mutex> lock(ctx->buf_mutex);
        if (ctx->buf_cached_bytes + cap <= kBufCacheLimit && ctx->buf_free.size() < 256) {
            ctx->buf_free.emplace_back(p, cap);
            ctx->buf_cached_bytes += cap;
            return;
        }
    }
    (void)hipFree(p);
}

fa_status ensure_scratch(fa_ctx *ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return FA_SUCCESS;
    if (ctx->scratch) {
        FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    FA_HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return FA_SUCCESS;
}
}  // namespace fa

extern "C" {

const char *fa_version(void) { return "fluidaudio_hip 0.1 gfx950"; }

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
    if (const char *lim = getenv("FLUIDAUDIO_HIP_WORKSPACE_LIMIT")) {
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
    if (ctx->mel_cache && ctx->mel_cache_free) ctx->mel_cache_free(ctx->mel_cache);
    for (auto &e : ctx->ahc_ev) if (e) (void)hipEventDestroy(e);
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
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    ctx->ws_limit = bytes;
    if (!ctx->ws_busy && ctx->ahc_ws && ctx->ahc_ws_bytes > bytes) free_caches_locked(ctx, false);
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
    std::lock_guard<std::mutex> lock(g_registry_mutex);
    if (ctx->ws_busy) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "trim: the context is inside a linkage call");
    free_caches_locked(ctx, true);
    { fa::DeviceGuard guard(ctx->device); (void)hipStreamSynchronize(ctx->stream); std::lock_guard<std::mutex> bl(ctx->buf_mutex); fa::buf_cache_flush(ctx); }
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
