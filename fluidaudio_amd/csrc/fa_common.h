// fa_common.h — shared plumbing of libfluidaudio_hip.so: context, error capture, workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/fluidaudio_hip.h"

struct fa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::string last_error;
    // grow-only device scratch (reused across calls so steady-state calls do not hipMalloc)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // AHC workspace cache (N^2 fp64 matrix etc.)
    void *ahc_ws = nullptr;
    size_t ahc_ws_bytes = 0;
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

// RAII device buffer for one-call temporaries on the host-pointer entry points.
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

fa_status ensure_scratch(fa_ctx *ctx, size_t bytes);

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace fa
