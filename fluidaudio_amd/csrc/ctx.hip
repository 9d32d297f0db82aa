// ctx.hip — context lifetime for libfluidaudio_hip.so.
#include "fa_common.h"

namespace fa {
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
    *out = ctx;
    return FA_SUCCESS;
}

void fa_ctx_destroy(fa_ctx *ctx) {
    if (!ctx) return;
    fa::DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->ahc_ws) (void)hipFree(ctx->ahc_ws);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
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
