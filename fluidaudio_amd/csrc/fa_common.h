// fa_common.h — shared plumbing of libfluidaudio_hip.so: context, error capture, workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/fluidaudio_hip.h"

struct fa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    // grow-only device scratch (reused across calls so steady-state calls do not hipMalloc)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // AHC workspace cache (N^2 fp64 matrix etc.): kept between calls (the first call at a new size pays 0.4 - 2.5 s of hipMalloc),
    // released when it exceeds ws_limit after a call, by fa_ctx_trim, or — while no call is using it (ws_busy, guarded by the
    // registry of ctx.hip) — by ANOTHER context of the same device whose own allocation failed.
    void *ahc_ws = nullptr;
    size_t ahc_ws_bytes = 0;
    size_t ws_limit = static_cast<size_t>(-1);   // bytes a context may keep cached between calls (FLUIDAUDIO_HIP_WORKSPACE_LIMIT)
    size_t ws_cap = static_cast<size_t>(-1);     // a linkage call needing more workspace than this fails with ALLOCATION_FAILURE
    bool ws_busy = false;
    // linkage: per-call set-up that does not depend on the data is kept (events; the captured graph of round launches, valid as long as
    // the workspace address and the problem shape are the same) — a short recording is dominated by such fixed costs
    hipEvent_t ahc_ev[3] = {nullptr, nullptr, nullptr};
    void *ahc_graph = nullptr;                 // owned by ahc.hip (ahc_graph_free releases it)
    void (*ahc_graph_free)(void *) = nullptr;
    void *ahc_uni_graph = nullptr;             // the same for the round launches of a uniform batch (ahc_batch_uniform): a batch job repeats one shape
    // linkage batches of a few LARGE problems run their merge chains concurrently, one per helper context (own stream, own workspace): see
    // ahc_run_device_batch.  Created on first use, trimmed and destroyed with this context.
    fa_ctx *helpers[3] = {nullptr, nullptr, nullptr};
    // mel: plans of small host-pointer calls (tables + geometry on the device) are kept (mel.hip) — a streaming caller repeats one shape
    void *mel_cache = nullptr;
    void (*mel_cache_free)(void *) = nullptr;
    // polyphase resampler taps of the last (up, down) pair, device resident (resample.hip)
    void *poly_taps = nullptr;
    size_t poly_taps_bytes = 0;
    int32_t poly_up = 0, poly_down = 0, poly_half = 0;
    void *poly_rows = nullptr;                 // per-phase tables of the same pair for poly_rows_kernel (owned by resample.hip)
    void (*poly_rows_free)(void *) = nullptr;
    // Device buffers of the clustering stage (inputs, VBx state, centroids, scores: ~30 per recording) are kept by the context that allocated
    // them and handed out again (fa::DevBuf::alloc(ctx, bytes), ctx.hip): a context is ONE stream, so reuse is ordered by the stream, and neither
    // the hipMalloc nor the hipFree — which waits for EVERY stream of the device, i.e. for the other recordings' kernels — is paid per call.
    std::mutex buf_mutex;                      // buffers may come back from another thread than the one using the context
    std::vector<std::pair<void *, size_t>> buf_free;
    size_t buf_cached_bytes = 0;
    // fa_ctx_set_timing: entries that support it bracket their DEVICE work (after their allocations) with two events on the stream, so a caller
    // can tell kernel time from host-side allocation time (bench.py's beam-search leg); last_device_ms < 0: nothing recorded
    bool timing = false;
    hipEvent_t tim_ev[2] = {nullptr, nullptr};
    double last_device_ms = -1.0;
    // fa_offline_cluster_batch prepares / finishes its recordings on worker contexts (own stream each); kept between calls since round 4
    fa_ctx *workers[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace fa {

inline fa_status set_error(fa_ctx *ctx, fa_status st, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->last_error = buf;
    }
    return st;
}

inline fa_status hip_status(fa_ctx *ctx, hipError_t e, const char *what) {
    if (e == hipSuccess) return FA_SUCCESS;
    const fa_status st = (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) ? FA_ALLOCATION_FAILURE : FA_RUNTIME_ERROR;
    return set_error(ctx, st, "%s: %s", what, hipGetErrorString(e));
}

// No exception crosses the C ABI (the reference's wrapper: std::bad_alloc -> ALLOCATION_FAILURE, anything else -> UNKNOWN_ERROR,
// FastClusterWrapper.cpp:236-243).  For entry points whose host side allocates (std::vector / std::string).
template <class F>
inline fa_status no_throw(fa_ctx *ctx, const char *what, F &&f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        try { return set_error(ctx, FA_ALLOCATION_FAILURE, "%s: host allocation failed", what); } catch (...) { return FA_ALLOCATION_FAILURE; }
    } catch (...) {
        try { return set_error(ctx, FA_UNKNOWN_ERROR, "%s: unexpected failure", what); } catch (...) { return FA_UNKNOWN_ERROR; }
    }
}

#define FA_HIP_TRY(ctx, expr)                                              \
    do {                                                                   \
        const hipError_t fa_e_ = (expr);                                   \
        if (fa_e_ != hipSuccess) return ::fa::hip_status((ctx), fa_e_, #expr); \
    } while (0)

#define FA_TRY(expr)                            \
    do {                                        \
        const fa_status fa_s_ = (expr);         \
        if (fa_s_ != FA_SUCCESS) return fa_s_;  \
    } while (0)

// Test hook (fa_debug_inject_fault, ctx.hip): true for the next `count` passes through `site`.  One relaxed atomic load when nothing is armed.
bool fault_hit(int site);

// ---- switches: every environment variable the library looks at, in ONE table (ctx.hip).  The environment is read ONCE per process (the first
// lookup; getenv is not safe against a concurrent setenv, and several host threads of one call dispatch at the same time); afterwards a value
// changes only through fa_debug_set_switch (tests; include/fluidaudio_hip.h documents the ROUTE switches and the hook).  A lookup is one
// atomic pointer load.  The AB switches select kernels / parameters for A/B measurements only: without -DFA_AB_SWITCHES (the release
// build) fa::sw() of one of them is the constant nullptr, so the code behind it is dead and dropped by the compiler.
#define FA_SWITCHES(ROUTE, AB)                                                                                                       \
    ROUTE(HIP_WORKSPACE_LIMIT, "FLUIDAUDIO_HIP_WORKSPACE_LIMIT") ROUTE(HIP_DEVICES, "FLUIDAUDIO_HIP_DEVICES") ROUTE(HIP_DEVICE, "FLUIDAUDIO_HIP_DEVICE") \
    ROUTE(AHC_CPT, "FA_AHC_CPT") ROUTE(AHC_NO_SINGLE_BLOCK, "FA_AHC_NO_SINGLE_BLOCK") ROUTE(AHC_NO_UNIFORM, "FA_AHC_NO_UNIFORM")      \
    ROUTE(AHC_RO_NO_MATRIX, "FA_AHC_RO_NO_MATRIX") ROUTE(AHC_RO_NO_HANDOVER, "FA_AHC_RO_NO_HANDOVER") ROUTE(AHC_UNI_CPT, "FA_AHC_UNI_CPT") ROUTE(AHC_UNI_GROUPS, "FA_AHC_UNI_GROUPS")    \
    ROUTE(AHC_UNI_WAVES, "FA_AHC_UNI_WAVES") ROUTE(AHC_IN_FLIGHT, "FA_AHC_IN_FLIGHT") ROUTE(AHC_DEBUG, "FA_AHC_DEBUG")                \
    ROUTE(MEL_GENERIC, "FA_MEL_GENERIC") ROUTE(MEL_SLICE_MB, "FA_MEL_SLICE_MB") ROUTE(VBX_NO_TILED, "FA_VBX_NO_TILED")                \
    ROUTE(RESAMPLE_SIMPLE, "FA_RESAMPLE_SIMPLE") ROUTE(RESAMPLE_NO_DECIM, "FA_RESAMPLE_NO_DECIM")                                     \
    ROUTE(RESAMPLE_NO_DECIM_TILES, "FA_RESAMPLE_NO_DECIM_TILES") ROUTE(RESAMPLE_NO_ROWS, "FA_RESAMPLE_NO_ROWS")                       \
    ROUTE(RESAMPLE_NO_WIDE, "FA_RESAMPLE_NO_WIDE") ROUTE(RESAMPLE_WIDE, "FA_RESAMPLE_WIDE")                                           \
    AB(AHC_GRAM_V1, "FA_AHC_GRAM_V1") AB(AHC_NO_MATRIX_FREE, "FA_AHC_NO_MATRIX_FREE") AB(AHC_ROM_DIRECT_START, "FA_AHC_ROM_DIRECT_START") \
    AB(AHC_ROUND_BIG, "FA_AHC_ROUND_BIG") AB(BEAM_PROF, "FA_BEAM_PROF") AB(CENTROID_SIMPLE, "FA_CENTROID_SIMPLE")                     \
    AB(MEL_NO_EZ, "FA_MEL_NO_EZ") AB(MEL_PRIO, "FA_MEL_PRIO") AB(MEL_PROF, "FA_MEL_PROF") AB(MEL_ROUNDS, "FA_MEL_ROUNDS")             \
    AB(MEL_SCALAR, "FA_MEL_SCALAR") AB(MEL_V4, "FA_MEL_V4") AB(MEL_V4_DEEP, "FA_MEL_V4_DEEP") AB(MEL_V4_LDS_PAD, "FA_MEL_V4_LDS_PAD") \
    AB(RESAMPLE_NO_INTERP, "FA_RESAMPLE_NO_INTERP") AB(RESAMPLE_ROWS_LDS_KB, "FA_RESAMPLE_ROWS_LDS_KB")                               \
    AB(RESAMPLE_ROWS_SHARE, "FA_RESAMPLE_ROWS_SHARE") AB(RESAMPLE_WIDE_NO_ROT, "FA_RESAMPLE_WIDE_NO_ROT")                             \
    AB(RESAMPLE_WIDE_PART, "FA_RESAMPLE_WIDE_PART")
enum class Sw : int {
#define FA_SW_ENUM(id, name) id,
    FA_SWITCHES(FA_SW_ENUM, FA_SW_ENUM)
#undef FA_SW_ENUM
    kCount
};
constexpr bool sw_is_ab(const Sw s) {
#define FA_SW_NO(id, name) if (s == Sw::id) return false;
#define FA_SW_YES(id, name) if (s == Sw::id) return true;
    FA_SWITCHES(FA_SW_NO, FA_SW_YES)
#undef FA_SW_NO
#undef FA_SW_YES
    return false;
}
const char *sw_lookup(Sw s);   // ctx.hip: the value (a string that lives for the whole process) or nullptr
__attribute__((always_inline)) inline const char *sw(const Sw s) {
#ifndef FA_AB_SWITCHES
    if (sw_is_ab(s)) return nullptr;
#endif
    return sw_lookup(s);
}
inline bool sw_on(const Sw s) { return sw(s) != nullptr; }

// A host thread for `f`, appended to `pool`; false when none is to be had (std::system_error from the constructor, std::bad_alloc from the
// vector, or an armed FA_FAULT_THREAD_START) — the caller then runs that share itself.  An exception escaping while `pool` holds joinable threads
// would std::terminate the process across the C ABI.
template <class F>
inline bool start_thread(std::vector<std::thread> &pool, F &&f) noexcept {
    if (fault_hit(FA_FAULT_THREAD_START)) return false;
    try {
        pool.emplace_back(std::forward<F>(f));
        return true;
    } catch (...) {
        return false;
    }
}

// RAII device buffer.  alloc(bytes): a one-call temporary of the host-pointer entry points (hipMalloc / hipFree).  alloc(ctx, bytes): taken
// from / returned to the context's buffer cache (ctx.hip) — for buffers used on that context's stream only.
hipError_t devbuf_take(fa_ctx *ctx, size_t bytes, void **p, size_t *cap);
void devbuf_give(fa_ctx *ctx, void *p, size_t cap);
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    fa_ctx *owner = nullptr;
    ~DevBuf() { reset(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;             // one owner: a copy would release the buffer twice
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), cap(o.cap), owner(o.owner) { o.p = nullptr; o.cap = 0; o.owner = nullptr; }
    DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { reset(); p = o.p; cap = o.cap; owner = o.owner; o.p = nullptr; o.cap = 0; o.owner = nullptr; } return *this; }
    void reset() { if (p) { if (owner) devbuf_give(owner, p, cap); else (void)hipFree(p); } p = nullptr; cap = 0; owner = nullptr; }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    hipError_t alloc(fa_ctx *ctx, size_t bytes) {
        const hipError_t e = devbuf_take(ctx, bytes ? bytes : 1, &p, &cap);
        if (e == hipSuccess) owner = ctx;
        return e;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

fa_status ensure_scratch(fa_ctx *ctx, size_t bytes);

// The cached linkage workspace of a context (ctx.hip).  ws_acquire marks it in use and makes it at least `bytes` large: a failed
// hipMalloc first releases the idle caches (scratch + linkage workspaces) of every OTHER context on the same device and retries, so a
// pool of contexts on one GPU degrades to re-allocating instead of failing; ws_release ends the use and applies ctx->ws_limit.
fa_status ws_acquire(fa_ctx *ctx, size_t bytes);
void ws_release(fa_ctx *ctx);
struct WsUse {   // scope of one linkage call
    fa_ctx *ctx;
    explicit WsUse(fa_ctx *c) : ctx(c) {}
    ~WsUse() { ws_release(ctx); }
};

// ---- device-level cores shared by the host-pointer entries and fa_offline_cluster (internal, not part of the C ABI).
// All pointers prefixed d_ are DEVICE pointers; everything is enqueued on ctx->stream; results stay on the device.
// d_Z: device pointer, or (z_on_host) the caller's host buffer the dendrogram is copied to directly
fa_status ahc_run_device(fa_ctx *ctx, const double *d_data, size_t N, size_t d, double *d_Z, int mode, fa_ahc_stats *stats, bool z_on_host = false);   // ahc.hip
fa_status ahc_normalize_dev(fa_ctx *ctx, const double *d_x, double *d_out, int64_t n, int32_t d);                              // ahc.hip (:70-105)
// count independent problems advanced by the same round launches; d_data / d_Z: HOST arrays of device pointers; statuses, stats: per problem (nullable)
fa_status ahc_run_device_batch(fa_ctx *ctx, int count, const double *const *d_data, const size_t *n, size_t d, double *const *d_Z, int mode,
                               fa_ahc_stats *stats, fa_status *statuses);

struct VbxDevice {   // buffers of one VBx run; gamma [T][S], pi [S] and hard [T] stay on the device for the stages behind it
    DevBuf phi, rho, G, gamma, pi, logpi, part, alpha, invL, phiT, ll, scal, hard;
    int64_t T = 0;
    int32_t D = 0, S = 0;
};
fa_status vbx_run_dev(fa_ctx *ctx, const double *d_X, int64_t T, int32_t D, const int32_t *d_labels, int32_t S, const double *phi_host,
                      double Fa, double Fb, int32_t max_iter, double epsilon, double *elbos_host, int32_t *n_iters, VbxDevice &out);   // vbx.hip

// VBxClustering.refine's catch block (VBxClustering.swift:136-141) on the device: gamma = one-hot labels, pi = 1/S, hard = clamped labels
fa_status vbx_degrade_dev(fa_ctx *ctx, int64_t T, int32_t S, const int32_t *d_labels, VbxDevice &out);                             // vbx.hip

fa_status centroids_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_gamma, int32_t S, const int32_t *d_spk, int32_t K,
                        double *d_cent, bool rows_finite = false);                                                                                             // post.hip
fa_status scores_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_cent, int32_t K, double *d_cn, double *d_scores);
fa_status assign_dev(fa_ctx *ctx, const double *d_emb, int64_t n, int32_t d, const double *d_cent, int32_t K, double *d_cn, int32_t *d_out);
fa_status constrained_assign_dev(fa_ctx *ctx, const double *d_scores, int64_t n, int32_t K, const int32_t *chunk_indices_host, int32_t *d_out);

fa_status default_pool(fa_pool **out);   // pool.hip: the device set behind the context-free drop-in symbol

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace fa
