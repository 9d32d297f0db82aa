// pool.hip — a device set for single-process hosts.
//
// The reference has no multi-device notion at all (no collectives, every manager owns its own CoreML models), and a Swift
// host linking this library gets no torch.distributed.  What the hot path needs across GPUs is only what SURVEY §8e states:
// utterances / logit matrices / recordings are independent units, so a host shards them contiguously and nothing is exchanged.
// This file gives the C ABI exactly that:
//   * fa_pool: one context (own stream, own workspace cache) per listed device, handed out to concurrent callers
//     (fa_pool_acquire / fa_pool_release); the drop-in symbol fastcluster_compute_centroid_linkage draws from a default pool, so
//     concurrent AHCClustering.cluster calls (OfflineDiarizerManager.swift:270 from several managers) land on different GPUs
//     instead of queueing on one (FLUIDAUDIO_HIP_DEVICES="0,1,..", default: every visible device);
//   * fa_mel_batch_sharded / fa_ctc_greedy_batch_sharded / fa_ahc_linkage_many: the host-pointer entries with the batch split
//     across the pool's contexts, one host thread per context, results written straight into the caller's buffers.
// The same device may be listed more than once ("0,0"): two contexts = two streams on one GPU, which is also how the 1-GPU test
// box exercises the sharding logic.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "fa_common.h"

struct fa_pool {
    std::vector<fa_ctx *> ctx;
    std::vector<char> busy;
    std::mutex m;
    std::condition_variable cv;
};

namespace {

bool parse_device_list(const char *s, std::vector<int> &out) {
    out.clear();
    if (!s) return false;
    const char *p = s;
    while (*p) {
        while (*p == ' ' || *p == ',') ++p;
        if (!*p) break;
        char *end = nullptr;
        const long v = strtol(p, &end, 10);
        if (end == p || v < 0 || v > 1 << 20) return false;
        out.push_back(static_cast<int>(v));
        p = end;
    }
    return !out.empty();
}

// contiguous ranges of `batch` units with (nearly) equal total weight; weight(b) >= 0
template <class W>
std::vector<int32_t> split_by_weight(int32_t batch, int parts, W weight) {
    std::vector<int32_t> cut(static_cast<size_t>(parts) + 1, batch);
    cut[0] = 0;
    double total = 0.0;
    for (int32_t b = 0; b < batch; ++b) total += static_cast<double>(weight(b));
    double acc = 0.0;
    int p = 1;
    for (int32_t b = 0; b < batch && p < parts; ++b) {
        acc += static_cast<double>(weight(b));
        while (p < parts && acc >= total * p / parts) cut[static_cast<size_t>(p++)] = b + 1;
    }
    for (int q = 1; q <= parts; ++q) if (cut[static_cast<size_t>(q)] < cut[static_cast<size_t>(q) - 1]) cut[static_cast<size_t>(q)] = cut[static_cast<size_t>(q) - 1];
    cut[static_cast<size_t>(parts)] = batch;
    return cut;
}

// run job(i, ctx) for every context of the pool that has work, each on its own host thread; first failure wins
template <class Job>
fa_status run_sharded(fa_pool *pool, const std::vector<int32_t> &cut, Job job) {
    const int parts = static_cast<int>(cut.size()) - 1;
    std::vector<fa_status> st(static_cast<size_t>(parts), FA_SUCCESS);
    std::vector<std::thread> th;
    int last = -1;
    for (int i = 0; i < parts; ++i) if (cut[static_cast<size_t>(i) + 1] > cut[static_cast<size_t>(i)]) last = i;
    for (int i = 0; i < parts; ++i) {
        if (cut[static_cast<size_t>(i) + 1] <= cut[static_cast<size_t>(i)]) continue;
        auto body = [&, i]() {
            try { st[static_cast<size_t>(i)] = job(i, pool->ctx[static_cast<size_t>(i)]); }
            catch (const std::bad_alloc &) { st[static_cast<size_t>(i)] = FA_ALLOCATION_FAILURE; }
            catch (...) { st[static_cast<size_t>(i)] = FA_RUNTIME_ERROR; }
        };
        if (i == last) body();               // the calling thread takes the last shard
        else if (!fa::start_thread(th, body)) body();   // no host thread to be had: that shard runs here, before the next one starts
    }
    for (auto &t : th) t.join();
    for (const fa_status s : st) if (s != FA_SUCCESS) return s;
    return FA_SUCCESS;
}

// every context of the pool is taken for the duration of a sharded call (callers of fa_pool_acquire wait meanwhile)
struct WholePool {
    fa_pool *pool;
    explicit WholePool(fa_pool *p) : pool(p) {
        std::unique_lock<std::mutex> lock(pool->m);
        pool->cv.wait(lock, [&] { for (const char b : pool->busy) if (b) return false; return true; });
        for (char &b : pool->busy) b = 1;
    }
    ~WholePool() {
        { std::lock_guard<std::mutex> lock(pool->m); for (char &b : pool->busy) b = 0; }
        pool->cv.notify_all();
    }
};

std::mutex g_default_pool_mutex;
fa_pool *g_default_pool = nullptr;

}  // namespace

namespace fa {
fa_status default_pool(fa_pool **out) {
    std::lock_guard<std::mutex> lock(g_default_pool_mutex);
    if (!g_default_pool) {
        std::vector<int> devs;
        if (!parse_device_list(fa::sw(fa::Sw::HIP_DEVICES), devs)) {
            devs.clear();
            if (const char *one = fa::sw(fa::Sw::HIP_DEVICE)) devs.push_back(atoi(one));   // the round-1 variable: one device
        }
        const fa_status st = fa_pool_create(devs.empty() ? nullptr : devs.data(), static_cast<int32_t>(devs.size()), &g_default_pool);
        if (st != FA_SUCCESS) return st;
    }
    *out = g_default_pool;
    return FA_SUCCESS;
}
}  // namespace fa

extern "C" {

fa_status fa_device_count(int32_t *count) {
    if (!count) return FA_INVALID_ARGUMENT;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); *count = 0; return FA_RUNTIME_ERROR; }
    *count = c;
    return FA_SUCCESS;
}

fa_status fa_pool_create(const int32_t *devices, int32_t n_devices, fa_pool **out) {
    if (!out || n_devices < 0 || (n_devices > 0 && !devices)) return FA_INVALID_ARGUMENT;
    *out = nullptr;
    return fa::no_throw(nullptr, "pool create", [&]() -> fa_status {
    std::vector<int> devs;
    if (n_devices == 0) {
        int32_t c = 0;
        FA_TRY(fa_device_count(&c));
        if (c <= 0) return FA_RUNTIME_ERROR;   // no GPU: fail loudly
        for (int i = 0; i < c; ++i) devs.push_back(i);
    } else {
        devs.assign(devices, devices + n_devices);
    }
    fa_pool *pool = new (std::nothrow) fa_pool();
    if (!pool) return FA_ALLOCATION_FAILURE;
    for (const int dev : devs) {
        fa_ctx *ctx = nullptr;
        const fa_status st = fa_ctx_create(dev, nullptr, &ctx);
        if (st != FA_SUCCESS) { fa_pool_destroy(pool); return st; }
        pool->ctx.push_back(ctx);
    }
    pool->busy.assign(pool->ctx.size(), 0);
    *out = pool;
    return FA_SUCCESS;
    });
}

void fa_pool_destroy(fa_pool *pool) {
    if (!pool) return;
    for (fa_ctx *c : pool->ctx) fa_ctx_destroy(c);
    delete pool;
}

int32_t fa_pool_size(const fa_pool *pool) { return pool ? static_cast<int32_t>(pool->ctx.size()) : 0; }

fa_ctx *fa_pool_context(fa_pool *pool, int32_t index) {
    if (!pool || index < 0 || index >= static_cast<int32_t>(pool->ctx.size())) return nullptr;
    return pool->ctx[static_cast<size_t>(index)];
}

int32_t fa_ctx_device(const fa_ctx *ctx) { return ctx ? ctx->device : -1; }

fa_status fa_pool_acquire(fa_pool *pool, fa_ctx **ctx) {
    if (!pool || !ctx || pool->ctx.empty()) return FA_INVALID_ARGUMENT;
    std::unique_lock<std::mutex> lock(pool->m);
    // The LOWEST free context: a serial caller stays on one device (its warm clocks, its cached linkage workspace — round 2 rotated
    // round-robin, so a serial Swift caller hopped to a cold GPU on every call and every device ended up caching its own N^2 workspace);
    // other devices are used only while the lower ones are busy, i.e. by concurrent callers.
    size_t found = pool->ctx.size();
    pool->cv.wait(lock, [&] {
        for (size_t i = 0; i < pool->ctx.size(); ++i)
            if (!pool->busy[i]) { found = i; return true; }
        return false;
    });
    pool->busy[found] = 1;
    *ctx = pool->ctx[found];
    return FA_SUCCESS;
}

void fa_pool_release(fa_pool *pool, fa_ctx *ctx) {
    if (!pool || !ctx) return;
    {
        std::lock_guard<std::mutex> lock(pool->m);
        for (size_t i = 0; i < pool->ctx.size(); ++i) if (pool->ctx[i] == ctx) pool->busy[i] = 0;
    }
    pool->cv.notify_all();
}

fa_status fa_mel_batch_sharded(fa_pool *pool, const fa_mel_config *cfg, const float *pcm, const int64_t *offsets, int32_t batch,
                               const float *last_samples, const int32_t *expected_frames, int32_t frame_stride, float *mel,
                               int32_t *mel_lengths) {
    if (!pool || pool->ctx.empty() || !cfg || !offsets || batch < 0 || (batch > 0 && (!pcm || !mel))) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    // the output geometry is that of the WHOLE batch: frame stride = the largest padded frame count (as fa_mel_plan_create), so
    // that utterance b lands at mel + b * n_mels * frame_stride whichever device computed it
    int32_t fs = frame_stride;
    if (fs <= 0) {
        fs = 1;
        for (int32_t b = 0; b < batch; ++b) {
            const int64_t len = offsets[b + 1] - offsets[b];
            if (len < 0) return FA_INVALID_ARGUMENT;
            int32_t T = fa_mel_num_frames(cfg, len);
            if (expected_frames && len > 0) T = expected_frames[b] > 0 ? expected_frames[b] : 0;   // expectedFrameCount (:347)
            const int32_t Tp = T > 0 ? fa_mel_padded_frames(cfg, T) : 1;
            if (Tp > fs) fs = Tp;
        }
    }
    const int64_t utt_stride = static_cast<int64_t>(cfg->n_mels) * fs;
    const auto cut = split_by_weight(batch, static_cast<int>(pool->ctx.size()), [&](int32_t b) { return offsets[b + 1] - offsets[b] + 1; });
    WholePool hold(pool);
    return run_sharded(pool, cut, [&](int i, fa_ctx *ctx) -> fa_status {
        const int32_t b0 = cut[static_cast<size_t>(i)], b1 = cut[static_cast<size_t>(i) + 1];
        std::vector<int64_t> off(static_cast<size_t>(b1 - b0) + 1);
        for (int32_t b = b0; b <= b1; ++b) off[static_cast<size_t>(b - b0)] = offsets[b] - offsets[b0];
        return fa_mel_batch(ctx, cfg, pcm + offsets[b0], off.data(), b1 - b0, last_samples ? last_samples + b0 : nullptr,
                            expected_frames ? expected_frames + b0 : nullptr, fs, mel + static_cast<int64_t>(b0) * utt_stride,
                            mel_lengths ? mel_lengths + b0 : nullptr);
    });
}

fa_status fa_ctc_greedy_batch_sharded(fa_pool *pool, const void *logits, int32_t dtype, int32_t batch, int32_t frames, int32_t vocab,
                                      int64_t row_stride, int64_t matrix_stride, const int32_t *valid_frames, int32_t blank_id,
                                      int32_t *frame_ids, int32_t *token_ids, int32_t *token_lens) {
    if (!pool || pool->ctx.empty() || batch < 0 || (batch > 0 && (!logits || !token_ids || !token_lens))) return FA_INVALID_ARGUMENT;
    if (batch == 0) return FA_SUCCESS;
    const size_t esz = dtype == FA_DTYPE_F16 ? 2 : 4;
    const auto cut = split_by_weight(batch, static_cast<int>(pool->ctx.size()), [&](int32_t) { return 1; });
    WholePool hold(pool);
    return run_sharded(pool, cut, [&](int i, fa_ctx *ctx) -> fa_status {
        const int32_t b0 = cut[static_cast<size_t>(i)], b1 = cut[static_cast<size_t>(i) + 1];
        const char *src = static_cast<const char *>(logits) + static_cast<size_t>(b0) * static_cast<size_t>(matrix_stride) * esz;
        return fa_ctc_greedy_batch(ctx, src, dtype, b1 - b0, frames, vocab, row_stride, matrix_stride, valid_frames ? valid_frames + b0 : nullptr,
                                   blank_id, frame_ids ? frame_ids + static_cast<int64_t>(b0) * frames : nullptr,
                                   token_ids + static_cast<int64_t>(b0) * frames, token_lens + b0);
    });
}

fa_status fa_ahc_linkage_many(fa_pool *pool, int32_t count, const double *const *data, const size_t *n, size_t d, double *const *dendrograms,
                              int32_t mode, fa_ahc_stats *stats, int32_t *statuses) {
    if (!pool || pool->ctx.empty() || count < 0 || (count > 0 && (!data || !n || !dendrograms))) return FA_INVALID_ARGUMENT;
    if (count == 0) return FA_SUCCESS;
    // recordings are dealt to the devices by merge-chain length (n), longest first, so that every device's batch finishes together
    const int parts = static_cast<int>(pool->ctx.size());
    std::vector<int32_t> order(static_cast<size_t>(count));
    for (int32_t i = 0; i < count; ++i) order[static_cast<size_t>(i)] = i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return n[a] > n[b]; });
    std::vector<std::vector<int32_t>> mine(static_cast<size_t>(parts));
    std::vector<double> load(static_cast<size_t>(parts), 0.0);
    for (const int32_t r : order) {
        int best = 0;
        for (int p = 1; p < parts; ++p) if (load[static_cast<size_t>(p)] < load[static_cast<size_t>(best)]) best = p;
        mine[static_cast<size_t>(best)].push_back(r);
        load[static_cast<size_t>(best)] += static_cast<double>(n[r]) + 1.0;
    }
    std::vector<int32_t> cut(static_cast<size_t>(parts) + 1, 0);   // run_sharded only needs "has work"
    for (int p = 0; p < parts; ++p) cut[static_cast<size_t>(p) + 1] = cut[static_cast<size_t>(p)] + static_cast<int32_t>(mine[static_cast<size_t>(p)].size());
    WholePool hold(pool);
    return run_sharded(pool, cut, [&](int i, fa_ctx *ctx) -> fa_status {
        const auto &ids = mine[static_cast<size_t>(i)];
        const int32_t k = static_cast<int32_t>(ids.size());
        std::vector<const double *> dp(ids.size());
        std::vector<double *> zp(ids.size());
        std::vector<size_t> np(ids.size());
        std::vector<fa_ahc_stats> sp(ids.size());
        std::vector<int32_t> stp(ids.size(), 0);
        for (size_t j = 0; j < ids.size(); ++j) { dp[j] = data[ids[j]]; zp[j] = dendrograms[ids[j]]; np[j] = n[ids[j]]; }
        const fa_status st = fa_ahc_linkage_batch(ctx, k, dp.data(), np.data(), d, zp.data(), mode, 0, stats ? sp.data() : nullptr, stp.data());
        for (size_t j = 0; j < ids.size(); ++j) {
            if (stats) stats[ids[j]] = sp[j];
            if (statuses) statuses[ids[j]] = stp[j];
        }
        return st;
    });
}

}  // extern "C"
