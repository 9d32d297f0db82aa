// fluidaudio.hpp — C++17 host-side mirror of the Swift types on FluidAudio's hot path, header-only, over the C ABI of
// include/fluidaudio_hip.h (libfluidaudio_hip.so).  Same names, argument meaning and error behaviour as the reference types
// (file:line cited per class), so that a C++ host — or a Swift shim generated from it — reads like the reference's callers.
// The reference's toolchain (Swift) is not in this image; its host language is compiled, hence this mirror is C++ (the
// Python package fluidaudio_amd/ is the ctypes twin used by the pytest suite).  No arithmetic lives here: guards that the
// reference performs before touching data are repeated so that the error behaviour is identical, everything else is a call.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "fluidaudio_hip.h"

namespace fluidaudio {

struct Error : std::runtime_error {
    fa_status status;
    Error(fa_status st, const std::string &where, const char *detail = nullptr)
        : std::runtime_error(where + ": status " + std::to_string(static_cast<int>(st)) + (detail && *detail ? std::string(" (") + detail + ")" : "")), status(st) {}
};

// One device context (stream + scratch).  The reference's value types are "one instance per thread" (AudioMelSpectrogram.swift:48-57);
// the same holds for a Context.
class Context {
public:
    explicit Context(int device = 0) { const fa_status st = fa_ctx_create(device, nullptr, &h_); if (st != FA_SUCCESS) throw Error(st, "fa_ctx_create"); }
    ~Context() { fa_ctx_destroy(h_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    fa_ctx *handle() const { return h_; }
    void check(fa_status st, const char *where) const { if (st != FA_SUCCESS) throw Error(st, where, fa_ctx_last_error(h_)); }
    // Memory policy of the cached linkage workspace (N^2 * 8 bytes: 15 GB at 43 200 embeddings): keep at most `bytes` between calls /
    // refuse calls that need more than `bytes` (ALLOCATION_FAILURE = the reference's status 4: AHCClustering degrades to singletons) /
    // release everything cached now / what is cached (helper contexts of in-flight batches included).
    void setWorkspaceLimit(size_t bytes) { check(fa_ctx_set_workspace_limit(h_, bytes), "fa_ctx_set_workspace_limit"); }
    void setWorkspaceCap(size_t bytes) { check(fa_ctx_set_workspace_cap(h_, bytes), "fa_ctx_set_workspace_cap"); }
    void trim() { check(fa_ctx_trim(h_), "fa_ctx_trim"); }
    size_t workspaceBytes() const { return fa_ctx_workspace_bytes(h_); }
    // server start-up: the workspace of `recordings` linkage problems of up to nMax embeddings now, not inside the first request
    void reserve(size_t nMax, size_t d, int32_t recordings = 1) { check(fa_ctx_reserve(h_, nMax, d, recordings), "fa_ctx_reserve"); }
private:
    fa_ctx *h_ = nullptr;
};

// A set of devices for one host process (fa_pool): `lease()` hands the calling thread a context of a free device for the
// lifetime of the returned object — what a host with several OfflineDiarizerManager instances (one per worker thread, the
// reference's model: OfflineDiarizerManager.swift:270) needs to use every GPU; the *_sharded calls split one batch over all of them.
class DevicePool {
public:
    DevicePool() { const fa_status st = fa_pool_create(nullptr, 0, &h_); if (st != FA_SUCCESS) throw Error(st, "fa_pool_create"); }
    explicit DevicePool(const std::vector<int32_t> &devices) {
        const fa_status st = fa_pool_create(devices.data(), static_cast<int32_t>(devices.size()), &h_);
        if (st != FA_SUCCESS) throw Error(st, "fa_pool_create");
    }
    ~DevicePool() { fa_pool_destroy(h_); }
    DevicePool(const DevicePool &) = delete;
    DevicePool &operator=(const DevicePool &) = delete;
    int size() const { return fa_pool_size(h_); }
    fa_pool *handle() const { return h_; }
    struct Lease {
        fa_pool *pool; fa_ctx *ctx;
        Lease(fa_pool *p, fa_ctx *c) : pool(p), ctx(c) {}
        Lease(Lease &&o) noexcept : pool(o.pool), ctx(o.ctx) { o.ctx = nullptr; }
        Lease(const Lease &) = delete;
        Lease &operator=(const Lease &) = delete;
        ~Lease() { if (ctx) fa_pool_release(pool, ctx); }
        int device() const { return fa_ctx_device(ctx); }
    };
    Lease lease() { fa_ctx *c = nullptr; const fa_status st = fa_pool_acquire(h_, &c); if (st != FA_SUCCESS) throw Error(st, "fa_pool_acquire"); return Lease(h_, c); }
    // recordings across the devices (fa_ahc_linkage_many): row-major fp64 matrices in, scipy-layout dendrograms out, per-problem statuses
    std::vector<fa_status> linkageMany(const std::vector<std::vector<double>> &data, size_t d, std::vector<std::vector<double>> &dendrograms) {
        const size_t k = data.size();
        std::vector<const double *> dp(k);
        std::vector<double *> zp(k);
        std::vector<size_t> n(k);
        std::vector<int32_t> st(k, 0);
        dendrograms.assign(k, {});
        static double dummy[4];
        for (size_t i = 0; i < k; ++i) {
            n[i] = d ? data[i].size() / d : 0;
            dendrograms[i].assign(n[i] > 1 ? (n[i] - 1) * 4 : 0, 0.0);
            dp[i] = data[i].empty() ? dummy : data[i].data();
            zp[i] = dendrograms[i].empty() ? dummy : dendrograms[i].data();
        }
        (void)fa_ahc_linkage_many(h_, static_cast<int32_t>(k), dp.data(), n.data(), d, zp.data(), FA_AHC_MODE_AUTO, nullptr, st.data());
        std::vector<fa_status> out(k);
        for (size_t i = 0; i < k; ++i) out[i] = static_cast<fa_status>(st[i]);
        return out;
    }
private:
    fa_pool *h_ = nullptr;
};

// ------------------------------------------------------------------------------------------------------------------ mel
// AudioMelSpectrogram (Sources/FluidAudio/Shared/AudioMelSpectrogram.swift:59-70 ctor, :185-292 computeFlat, :325-456 computeFlatTransposed)
class AudioMelSpectrogram {
public:
    enum class LogFloorMode { additive, clamped };   // :24-27
    enum class PaddingMode { center, prePadded };    // :19-22
    struct Flat { std::vector<float> mel; int melLength; int numFrames; };

    AudioMelSpectrogram(Context &ctx, int sampleRate = 16000, int nMels = 128, int nFFT = 512, int hopLength = 160, int winLength = 400,
                        float preemph = 0.97f, int padTo = 0, float logFloor = std::ldexp(1.0f, -24), LogFloorMode logFloorMode = LogFloorMode::additive,
                        bool windowPeriodic = false)
        : ctx_(ctx) {
        fa_mel_default_config(&cfg_);
        cfg_.sample_rate = sampleRate; cfg_.n_mels = nMels; cfg_.n_fft = nFFT; cfg_.hop = hopLength; cfg_.win = winLength; cfg_.preemph = preemph;
        cfg_.pad_to = padTo; cfg_.log_floor = logFloor; cfg_.floor_mode = logFloorMode == LogFloorMode::clamped ? FA_MEL_FLOOR_CLAMPED : FA_MEL_FLOOR_ADDITIVE;
        cfg_.window_periodic = windowPeriodic ? 1 : 0;
    }
    // computeFlat(audio:lastAudioSample:) -> (mel [nMels, numFrames] flat, melLength, numFrames)
    Flat computeFlat(const std::vector<float> &audio, float lastAudioSample = 0.0f) const { return run(audio, lastAudioSample, FA_MEL_PAD_CENTER, FA_MEL_LAYOUT_MEL_MAJOR, std::nullopt); }
    // computeFlatTransposed(audio:lastAudioSample:paddingMode:expectedFrameCount:) -> (mel [numFrames, nMels] flat, melLength, numFrames)
    Flat computeFlatTransposed(const std::vector<float> &audio, float lastAudioSample = 0.0f, PaddingMode paddingMode = PaddingMode::center,
                               std::optional<int> expectedFrameCount = std::nullopt) const {
        return run(audio, lastAudioSample, paddingMode == PaddingMode::prePadded ? FA_MEL_PAD_PREPADDED : FA_MEL_PAD_CENTER, FA_MEL_LAYOUT_FRAME_MAJOR, expectedFrameCount);
    }
    std::vector<float> hannWindow() const { std::vector<float> w(cfg_.win); fa_mel_hann_window(&cfg_, w.data()); return w; }
    std::vector<float> melFilterbankFlat() const { std::vector<float> f(static_cast<size_t>(cfg_.n_mels) * (cfg_.n_fft / 2 + 1)); fa_mel_filterbank(&cfg_, f.data()); return f; }

private:
    Flat run(const std::vector<float> &audio, float last, int pad, int layout, std::optional<int> expected) const {
        fa_mel_config c = cfg_;
        c.padding_mode = pad; c.layout = layout;
        int T = expected ? *expected : fa_mel_num_frames(&c, static_cast<int64_t>(audio.size()));
        if (T <= 0 || audio.empty()) return Flat{std::vector<float>(static_cast<size_t>(c.n_mels), 0.0f), 0, 1};   // :199-201, :349-351
        const int Tpad = fa_mel_padded_frames(&c, T);
        Flat out{std::vector<float>(static_cast<size_t>(c.n_mels) * Tpad, 0.0f), 0, Tpad};
        const int64_t offs[2] = {0, static_cast<int64_t>(audio.size())};
        int32_t len = 0, exp = T;
        ctx_.check(fa_mel_batch(ctx_.handle(), &c, audio.data(), offs, 1, &last, expected ? &exp : nullptr, Tpad, out.mel.data(), &len), "fa_mel_batch");
        out.melLength = len;
        return out;
    }
    Context &ctx_;
    fa_mel_config cfg_{};
};

// ------------------------------------------------------------------------------------------------------------------ CTC
using Vocabulary = std::map<int, std::string>;   // [Int: String]

// decodeCtcTokenIds (…/CTC/CtcDecoder.swift:289-294): pieces joined, U+2581 -> space, spaces trimmed
inline std::string decodeCtcTokenIds(const std::vector<int> &ids, const Vocabulary &vocabulary) {
    std::string s;
    for (int id : ids) { auto it = vocabulary.find(id); if (it != vocabulary.end()) s += it->second; }
    std::string o;
    for (size_t i = 0; i < s.size();) {
        if (s.compare(i, 3, "\xe2\x96\x81") == 0) { o += ' '; i += 3; } else o += s[i++];
    }
    const size_t a = o.find_first_not_of(" \t"), b = o.find_last_not_of(" \t");
    return a == std::string::npos ? std::string() : o.substr(a, b - a + 1);
}

// LogitsArgmax.argmaxPerFrame (Sources/FluidAudio/ASR/Shared/LogitsArgmax.swift:16-55) on a [frames, vocab] matrix with row stride
inline std::vector<int> argmaxPerFrame(Context &ctx, const float *logits, int frames, int vocab, int64_t rowStride) {
    std::vector<int32_t> ids(frames > 0 ? frames : 0), toks(ids.size());
    int32_t n = 0;
    if (frames > 0) ctx.check(fa_ctc_greedy_batch(ctx.handle(), logits, FA_DTYPE_F32, 1, frames, vocab, rowStride, static_cast<int64_t>(frames) * rowStride, nullptr, -1,
                                                   ids.data(), toks.data(), &n), "fa_ctc_greedy_batch");
    return std::vector<int>(ids.begin(), ids.end());
}

// ctcGreedyDecode(logProbs:vocabulary:blankId:) (CtcDecoder.swift:15-36): the [[Float]] overload — each frame's scan is seeded with frame[0]
// (a NaN there => index 0, :25), frames keep their own lengths (:26), empty frames are skipped before `prev` is touched (:23)
inline std::string ctcGreedyDecode(Context &ctx, const std::vector<std::vector<float>> &logProbs, const Vocabulary &vocabulary, int blankId = 1024) {
    const size_t T = logProbs.size();
    if (T == 0) return "";
    std::vector<int64_t> offs(T + 1, 0);
    for (size_t t = 0; t < T; ++t) offs[t + 1] = offs[t] + static_cast<int64_t>(logProbs[t].size());
    std::vector<float> flat;
    flat.reserve(static_cast<size_t>(offs[T]));
    for (const auto &r : logProbs) flat.insert(flat.end(), r.begin(), r.end());
    std::vector<int32_t> toks(T);
    int32_t n = 0;
    ctx.check(fa_ctc_greedy_rows(ctx.handle(), flat.data(), offs.data(), static_cast<int64_t>(T), nullptr, 1, blankId, nullptr, toks.data(), &n), "fa_ctc_greedy_rows");
    return decodeCtcTokenIds(std::vector<int>(toks.begin(), toks.begin() + n), vocabulary);
}

// ctcGreedyDecode(logProbs: MLMultiArray [1, T, V], …) (CtcDecoder.swift:45-70): contiguous rows, -inf seed (a NaN never wins)
inline std::string ctcGreedyDecode(Context &ctx, const float *logProbs, int T, int V, const Vocabulary &vocabulary, int blankId = 1024) {
    if (T <= 0 || V <= 0) return "";
    std::vector<int32_t> toks(T);
    int32_t n = 0;
    ctx.check(fa_ctc_greedy_batch(ctx.handle(), logProbs, FA_DTYPE_F32, 1, T, V, static_cast<int64_t>(V), static_cast<int64_t>(T) * V, nullptr, blankId, nullptr,
                                  toks.data(), &n), "fa_ctc_greedy_batch");
    return decodeCtcTokenIds(std::vector<int>(toks.begin(), toks.begin() + n), vocabulary);
}

// ARPALanguageModel (…/CTC/ARPALanguageModel.swift:16-104)
class ARPALanguageModel {
public:
    static constexpr float unkLogProb = -23.026f;
    explicit ARPALanguageModel(const std::string &arpaText) { const fa_status st = fa_arpa_parse(nullptr, arpaText.data(), static_cast<int64_t>(arpaText.size()), &h_); if (st != FA_SUCCESS) throw Error(st, "fa_arpa_parse"); }
    ~ARPALanguageModel() { fa_arpa_destroy(h_); }
    ARPALanguageModel(const ARPALanguageModel &) = delete;
    ARPALanguageModel &operator=(const ARPALanguageModel &) = delete;
    int64_t unigramCount() const { return fa_arpa_unigram_count(h_); }
    int64_t bigramContextCount() const { return fa_arpa_bigram_context_count(h_); }
    float score(const std::string &word, const std::optional<std::string> &prev = std::nullopt) const {   // :98-103
        float out = 0.0f;
        fa_arpa_score(h_, word.c_str(), prev ? prev->c_str() : nullptr, &out);
        return out;
    }
    fa_arpa_lm *handle() const { return h_; }
private:
    fa_arpa_lm *h_ = nullptr;
};

// ctcBeamSearch(logProbs:vocabulary:lm:beamWidth:lmWeight:wordBonus:blankId:tokenCandidates:) (CtcDecoder.swift:118-241)
inline std::string ctcBeamSearch(Context &ctx, const std::vector<std::vector<float>> &logProbs, const Vocabulary &vocabulary, const ARPALanguageModel *lm = nullptr,
                                 int beamWidth = 100, float lmWeight = 0.3f, float wordBonus = 0.0f, int blankId = 1024, int tokenCandidates = 40) {
    if (logProbs.empty() || logProbs[0].empty()) return "";   // :129-131
    const int T = static_cast<int>(logProbs.size()), V = static_cast<int>(logProbs[0].size());
    std::vector<float> flat;
    flat.reserve(static_cast<size_t>(T) * V);
    for (const auto &r : logProbs) flat.insert(flat.end(), r.begin(), r.begin() + V);
    fa_ctc_vocab *voc = nullptr;
    if (lm) {
        std::vector<int32_t> ids;
        std::vector<const char *> pieces;
        for (const auto &kv : vocabulary) { ids.push_back(kv.first); pieces.push_back(kv.second.c_str()); }
        ctx.check(fa_ctc_vocab_create(ctx.handle(), ids.data(), pieces.data(), static_cast<int32_t>(ids.size()), V, &voc), "fa_ctc_vocab_create");
    }
    std::vector<int32_t> toks(T);
    int32_t n = 0;
    float score = 0.0f;
    const fa_status st = fa_ctc_beam_search_batch(ctx.handle(), flat.data(), 1, T, V, nullptr, voc, lm ? lm->handle() : nullptr, beamWidth, lmWeight, wordBonus, blankId,
                                                  tokenCandidates, toks.data(), &n, &score);
    fa_ctc_vocab_destroy(voc);
    ctx.check(st, "fa_ctc_beam_search_batch");
    return decodeCtcTokenIds(std::vector<int>(toks.begin(), toks.begin() + n), vocabulary);
}

// ------------------------------------------------------------------------------------------------------------------ clustering
using Matrix = std::vector<std::vector<double>>;   // [[Double]]

inline std::vector<double> flatten(const Matrix &m, size_t &n, size_t &d) {
    n = m.size(); d = n ? m[0].size() : 0;
    std::vector<double> f;
    f.reserve(n * d);
    for (const auto &r : m) f.insert(f.end(), r.begin(), r.begin() + d);
    return f;
}

// AHCClustering.cluster(embeddingFeatures:threshold:) (Sources/FluidAudio/Diarizer/Offline/Clustering/AHCClustering.swift:20-67)
struct AHCClustering {
    Context &ctx;
    std::vector<int> cluster(const Matrix &embeddingFeatures, double threshold) const {
        size_t n, d;
        const std::vector<double> x = flatten(embeddingFeatures, n, d);
        if (n == 0) return {};                                   // :24
        if (d == 0) return std::vector<int>(n, 0);               // :25-27
        if (n == 1) return {0};                                  // :28
        std::vector<int32_t> labels(n);
        (void)fa_ahc_cluster(ctx.handle(), x.data(), n, d, threshold, FA_AHC_MODE_AUTO, labels.data(), nullptr);   // failure -> 0..<n inside (:52-55)
        return std::vector<int>(labels.begin(), labels.end());
    }
};

// SpeakerCountConstraints (…/Clustering/SpeakerCountConstraints.swift:6-77)
struct SpeakerCountConstraints {
    std::optional<int> numSpeakers;
    int minSpeakers, maxSpeakers;
    static SpeakerCountConstraints resolve(int numEmbeddings, std::optional<int> numSpeakers, std::optional<int> minSpeakers, std::optional<int> maxSpeakers) {
        int64_t a = numSpeakers.value_or(0), b = minSpeakers.value_or(0), c = maxSpeakers.value_or(0), out[3];
        fa_speaker_constraints_resolve(numEmbeddings, numSpeakers ? &a : nullptr, minSpeakers ? &b : nullptr, maxSpeakers ? &c : nullptr, out);
        return SpeakerCountConstraints{out[0] < 0 ? std::nullopt : std::optional<int>(static_cast<int>(out[0])), static_cast<int>(out[1]), static_cast<int>(out[2])};
    }
    bool needsAdjustment(int detectedCount) const { return detectedCount < minSpeakers || detectedCount > maxSpeakers; }
    int targetCount(int detectedCount) const { return detectedCount < minSpeakers ? minSpeakers : (detectedCount > maxSpeakers ? maxSpeakers : detectedCount); }
};

// KMeansClustering (…/Clustering/KMeansClustering.swift:39-129)
struct KMeansClustering {
    static std::pair<std::vector<int>, Matrix> clusterWithCentroids(Context &ctx, const Matrix &embeddings, int numClusters, int maxIterations = 300,
                                                                    std::optional<uint64_t> seed = std::nullopt, int nInit = 1) {
        size_t n, d;
        const std::vector<double> x = flatten(embeddings, n, d);
        if (n == 0) return {};
        std::vector<int32_t> labels(n);
        std::vector<double> cen(static_cast<size_t>(std::max<int64_t>(1, std::min<int64_t>(numClusters, static_cast<int64_t>(n)))) * std::max<size_t>(d, 1));
        int32_t k = 0;
        ctx.check(nInit > 1 ? fa_kmeans_cluster_ninit(ctx.handle(), x.data(), static_cast<int64_t>(n), static_cast<int32_t>(d), numClusters, maxIterations, nInit, seed.value_or(0),
                                                      labels.data(), cen.data(), &k, nullptr, nullptr)
                            : fa_kmeans_cluster(ctx.handle(), x.data(), static_cast<int64_t>(n), static_cast<int32_t>(d), numClusters, maxIterations, seed.value_or(0), labels.data(),
                                                cen.data(), &k, nullptr),
                  "fa_kmeans_cluster");
        Matrix c(k, std::vector<double>(d));
        for (int i = 0; i < k; ++i) std::copy(cen.begin() + static_cast<size_t>(i) * d, cen.begin() + static_cast<size_t>(i + 1) * d, c[i].begin());
        return {std::vector<int>(labels.begin(), labels.end()), c};
    }
    static std::pair<std::vector<int>, Matrix> clusterWithCentroidsNInit(Context &ctx, const Matrix &embeddings, int numClusters, int maxIterations = 300, int nInit = 10,
                                                                         uint64_t baseSeed = 0) {
        return clusterWithCentroids(ctx, embeddings, numClusters, maxIterations, baseSeed, nInit);
    }
};

// VBxOutput / VBxClustering.refine / refineWithConstraints (…/Clustering/VBxClustering.swift:41-165, :685-733; OfflineDiarizerTypes.swift:675-702)
struct VBxOutput {
    Matrix gamma;
    std::vector<double> pi;
    std::vector<int> hardClusters;
    Matrix centroids;
    int numClusters = 0;
    std::vector<double> elbos;
    bool wasAdjusted = false;
    std::optional<int> originalClusterCount;
    int assignedClusterCount() const {
        if (gamma.empty()) { int c = 0; for (double p : pi) c += p > 1e-7; return pi.empty() ? numClusters : c; }
        std::vector<char> win(gamma[0].size(), 0);
        for (const auto &row : gamma) { size_t b = 0; for (size_t i = 1; i < row.size(); ++i) if (row[i] > row[b]) b = i; if (!row.empty()) win[b] = 1; }
        int c = 0; for (char w : win) c += w; return c;
    }
};

struct VBxClustering {
    Context &ctx;
    std::vector<double> phiParameters;
    int maxIterations = 20;
    double convergenceTolerance = 1e-4, warmStartFa = 0.07, warmStartFb = 0.8;   // OfflineDiarizerTypes.swift:155-163,189-192

    VBxOutput refine(const Matrix &rhoFeatures, const std::vector<int> &initialClusters) const {
        size_t T, D;
        const std::vector<double> rho = flatten(rhoFeatures, T, D);
        if (T == 0 || D == 0) return {};                         // :45-67
        std::vector<int32_t> init(initialClusters.begin(), initialClusters.end());
        std::vector<double> phi = phiParameters.size() == D ? phiParameters : std::vector<double>(D, 1.0);   // :72-76
        const int S = std::max(1, fa_vbx_speaker_count(init.data(), static_cast<int64_t>(T)));
        std::vector<double> gamma(T * S), pi(S), elbos(std::max(maxIterations, 1));
        std::vector<int32_t> hard(T);
        int32_t it = 0, ns = 0;
        ctx.check(fa_vbx_refine(ctx.handle(), rho.data(), static_cast<int64_t>(T), static_cast<int32_t>(D), init.data(), phi.data(), warmStartFa, warmStartFb, maxIterations,
                                convergenceTolerance, gamma.data(), pi.data(), hard.data(), elbos.data(), &it, &ns), "fa_vbx_refine");
        VBxOutput o;
        o.gamma.assign(T, std::vector<double>(S));
        for (size_t t = 0; t < T; ++t) std::copy(gamma.begin() + t * S, gamma.begin() + (t + 1) * S, o.gamma[t].begin());
        o.pi = pi; o.hardClusters.assign(hard.begin(), hard.end()); o.numClusters = S; o.elbos.assign(elbos.begin(), elbos.begin() + it);
        return o;
    }
    VBxOutput refineWithConstraints(const Matrix &rhoFeatures, const Matrix &trainingEmbeddings, const std::vector<int> &initialClusters,
                                    const std::optional<SpeakerCountConstraints> &constraints) const {
        VBxOutput out = refine(rhoFeatures, initialClusters);
        if (!constraints) return out;
        const int detected = out.assignedClusterCount();
        if (!constraints->needsAdjustment(detected)) return out;
        const int target = constraints->targetCount(detected);
        auto km = KMeansClustering::clusterWithCentroidsNInit(ctx, trainingEmbeddings, target, 100, 10, 0);   // :716-722
        out.hardClusters = km.first; out.centroids = km.second; out.numClusters = target; out.wasAdjusted = true; out.originalClusterCount = detected;
        return out;
    }
};

// HungarianAssignment.maxScoreAssignment / ConstrainedClusterAssignment.assign (Diarizer/HungarianAssignment.swift:67-97,
// …/Clustering/ConstrainedClusterAssignment.swift:20-42)
struct ConstrainedClusterAssignment {
    static std::vector<int> assign(Context &ctx, const Matrix &scores, const std::vector<int> &chunkIndices) {
        size_t n, K;
        const std::vector<double> s = flatten(scores, n, K);
        std::vector<int32_t> chunks(chunkIndices.begin(), chunkIndices.end()), out(chunkIndices.size());
        if (!chunks.empty()) ctx.check(fa_constrained_assign(ctx.handle(), s.data(), static_cast<int64_t>(chunks.size()), static_cast<int32_t>(K), chunks.data(), out.data()), "fa_constrained_assign");
        return std::vector<int>(out.begin(), out.end());
    }
};

// OfflineDiarizerManager.cluster (…/Offline/Core/OfflineDiarizerManager.swift:270-375) on precomputed embeddings: one device-resident call
struct OfflineClusteringConfig {   // OfflineDiarizerTypes.swift:155-163,189-192
    double clusteringThreshold = 0.6, warmStartFa = 0.07, warmStartFb = 0.8, convergenceTolerance = 1e-4;
    int maxVbxIterations = 20;
    bool constrainedAssignment = true;
    std::optional<int> numSpeakers, minSpeakers, maxSpeakers;
    // the clustering / VBx guards of OfflineDiarizerConfig.validate (OfflineDiarizerTypes.swift:357-408: invalidConfiguration)
    void validate() const {
        if (!(clusteringThreshold > 0 && clusteringThreshold <= 2.0)) throw Error(FA_INVALID_ARGUMENT, "invalidConfiguration: clustering.threshold must be within (0, 2]");
        if (!(warmStartFa > 0 && warmStartFb > 0)) throw Error(FA_INVALID_ARGUMENT, "invalidConfiguration: clustering warm-start Fa/Fb must be positive");
        if (!(maxVbxIterations > 0)) throw Error(FA_INVALID_ARGUMENT, "invalidConfiguration: maxVBxIterations must be > 0");
        if (!(convergenceTolerance > 0)) throw Error(FA_INVALID_ARGUMENT, "invalidConfiguration: convergenceTolerance must be positive");
    }
};
struct OfflineClusteringResult {
    std::vector<int> assignments;   // -2: slot dropped by the constrained assignment
    Matrix centroids;
    fa_offline_cluster_info info{};
};
inline OfflineClusteringResult clusterEmbeddings(Context &ctx, const std::vector<std::vector<float>> &embeddings, const Matrix &rhoFeatures, const std::vector<int> &chunkIndices,
                                                 const std::vector<double> &phi, const OfflineClusteringConfig &config = {}) {
    config.validate();
    if (embeddings.empty()) throw Error(FA_INVALID_ARGUMENT, "noSpeechDetected");   // :281-283
    const size_t n = embeddings.size(), d = embeddings[0].size();
    std::vector<float> e(n * d);
    for (size_t i = 0; i < n; ++i) std::copy(embeddings[i].begin(), embeddings[i].end(), e.begin() + i * d);
    size_t rn = 0, rd = 0;
    const std::vector<double> rho = flatten(rhoFeatures, rn, rd);
    std::vector<double> ph = phi.size() == rd ? phi : std::vector<double>(rd, 1.0);   // VBxClustering.swift:72-76
    std::vector<int32_t> chunks(chunkIndices.begin(), chunkIndices.end()), labels(n);
    fa_offline_cluster_config c;
    fa_offline_cluster_default_config(&c);
    c.clustering_threshold = config.clusteringThreshold; c.warm_start_fa = config.warmStartFa; c.warm_start_fb = config.warmStartFb;
    c.max_vbx_iterations = config.maxVbxIterations; c.convergence_tolerance = config.convergenceTolerance; c.constrained_assignment = config.constrainedAssignment ? 1 : 0;
    c.num_speakers = config.numSpeakers.value_or(-1); c.min_speakers = config.minSpeakers.value_or(-1); c.max_speakers = config.maxSpeakers.value_or(-1);
    OfflineClusteringResult out;
    std::vector<double> cen(256 * d);
    int32_t k = 0;
    ctx.check(fa_offline_cluster(ctx.handle(), e.data(), static_cast<int64_t>(n), static_cast<int32_t>(d), rd ? rho.data() : nullptr, static_cast<int32_t>(rd), chunks.data(),
                                 rd ? ph.data() : nullptr, &c, 0, labels.data(), cen.data(), 256, &k, &out.info), "fa_offline_cluster");
    out.assignments.assign(labels.begin(), labels.end());
    out.centroids.assign(k, std::vector<double>(d));
    for (int i = 0; i < k; ++i) std::copy(cen.begin() + static_cast<size_t>(i) * d, cen.begin() + static_cast<size_t>(i + 1) * d, out.centroids[i].begin());
    return out;
}

// Several recordings through the stage in one call (fa_offline_cluster_batch: their merge chains advance together).  One entry per
// recording; a recording that fails (e.g. no embeddings) yields an empty optional and does not stop the others.
struct OfflineRecording { std::vector<std::vector<float>> embeddings; Matrix rhoFeatures; std::vector<int> chunkIndices; };
inline std::vector<std::optional<OfflineClusteringResult>> clusterEmbeddingsBatch(Context &ctx, const std::vector<OfflineRecording> &recordings, const std::vector<double> &phi,
                                                                                   const OfflineClusteringConfig &config = {}) {
    config.validate();
    const size_t count = recordings.size();
    std::vector<std::optional<OfflineClusteringResult>> out(count);
    if (count == 0) return out;
    size_t d = 0, rd = 0;
    for (const auto &r : recordings) { if (!r.embeddings.empty()) d = r.embeddings[0].size(); if (!r.rhoFeatures.empty()) rd = r.rhoFeatures[0].size(); }
    if (d == 0) return out;
    std::vector<std::vector<float>> e(count);
    std::vector<std::vector<double>> rho(count), cen(count, std::vector<double>(256 * d));
    std::vector<std::vector<int32_t>> chunks(count), labels(count);
    std::vector<const float *> ep(count);
    std::vector<const double *> rp(count);
    std::vector<const int32_t *> cp(count);
    std::vector<int32_t *> lp(count);
    std::vector<double *> zp(count);
    std::vector<int64_t> n(count);
    static const float dummy_f[4] = {0, 0, 0, 0};
    static const double dummy_d[4] = {0, 0, 0, 0};
    static const int32_t dummy_i[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < count; ++i) {
        const auto &r = recordings[i];
        n[i] = static_cast<int64_t>(r.embeddings.size());
        e[i].resize(r.embeddings.size() * d);
        for (size_t t = 0; t < r.embeddings.size(); ++t) std::copy(r.embeddings[t].begin(), r.embeddings[t].end(), e[i].begin() + t * d);
        size_t rn = 0, rdi = 0;
        rho[i] = flatten(r.rhoFeatures, rn, rdi);
        chunks[i].assign(r.chunkIndices.begin(), r.chunkIndices.end());
        labels[i].assign(std::max<size_t>(r.embeddings.size(), 1), 0);
        ep[i] = e[i].empty() ? dummy_f : e[i].data();
        rp[i] = rho[i].empty() ? dummy_d : rho[i].data();
        cp[i] = chunks[i].empty() ? dummy_i : chunks[i].data();
        lp[i] = labels[i].data();
        zp[i] = cen[i].data();
    }
    std::vector<double> ph = phi.size() == rd ? phi : std::vector<double>(rd, 1.0);
    fa_offline_cluster_config c;
    fa_offline_cluster_default_config(&c);
    c.clustering_threshold = config.clusteringThreshold; c.warm_start_fa = config.warmStartFa; c.warm_start_fb = config.warmStartFb;
    c.max_vbx_iterations = config.maxVbxIterations; c.convergence_tolerance = config.convergenceTolerance; c.constrained_assignment = config.constrainedAssignment ? 1 : 0;
    c.num_speakers = config.numSpeakers.value_or(-1); c.min_speakers = config.minSpeakers.value_or(-1); c.max_speakers = config.maxSpeakers.value_or(-1);
    std::vector<int32_t> k(count, 0), st(count, 0);
    std::vector<fa_offline_cluster_info> infos(count);
    (void)fa_offline_cluster_batch(ctx.handle(), static_cast<int32_t>(count), ep.data(), n.data(), static_cast<int32_t>(d), rd ? rp.data() : nullptr, static_cast<int32_t>(rd), cp.data(),
                                   rd ? ph.data() : nullptr, &c, lp.data(), zp.data(), 256, k.data(), infos.data(), st.data());
    for (size_t i = 0; i < count; ++i) {
        if (st[i] != FA_SUCCESS) continue;
        OfflineClusteringResult r;
        r.assignments.assign(labels[i].begin(), labels[i].begin() + n[i]);
        r.centroids.assign(k[i], std::vector<double>(d));
        for (int q = 0; q < k[i]; ++q) std::copy(cen[i].begin() + static_cast<size_t>(q) * d, cen[i].begin() + static_cast<size_t>(q + 1) * d, r.centroids[q].begin());
        r.info = infos[i];
        out[i] = std::move(r);
    }
    return out;
}

// LuxTtsMelExtractor (Sources/FluidAudio/TTS/LuxTts/LuxTtsMelExtractor.swift:15-132): the torchaudio-flavoured front end, same C ABI
struct LuxTtsMelExtractor {
    Context &ctx;
    static constexpr int nFFT = 1024, hop = 256, nMels = 100, sampleRate = 24000;
    int frameCount(int sampleCount) const { return (sampleCount + hop / 2) / hop; }   // :45-47
    std::vector<std::vector<float>> extract(const std::vector<float> &audio) const {
        const int T = frameCount(static_cast<int>(audio.size()));
        if (audio.empty() || T <= 0) return {};
        fa_mel_config c;
        fa_mel_default_config(&c);
        c.sample_rate = sampleRate; c.n_mels = nMels; c.n_fft = nFFT; c.hop = hop; c.win = nFFT; c.preemph = 0.0f; c.log_floor = 1e-7f; c.floor_mode = FA_MEL_FLOOR_CLAMPED;
        c.window_periodic = 1; c.layout = FA_MEL_LAYOUT_FRAME_MAJOR; c.power = 1.0f; c.center_pad = FA_MEL_CENTER_REFLECT; c.mel_scale = FA_MEL_SCALE_HTK_NONORM;
        c.tail_mode = FA_MEL_TAIL_REPLICATE;
        std::vector<float> flat(static_cast<size_t>(T) * nMels);
        const int64_t offs[2] = {0, static_cast<int64_t>(audio.size())};
        int32_t len = 0, exp = T;
        ctx.check(fa_mel_batch(ctx.handle(), &c, audio.data(), offs, 1, nullptr, &exp, T, flat.data(), &len), "fa_mel_batch");
        std::vector<std::vector<float>> out(T, std::vector<float>(nMels));
        for (int t = 0; t < T; ++t) std::copy(flat.begin() + static_cast<size_t>(t) * nMels, flat.begin() + static_cast<size_t>(t + 1) * nMels, out[t].begin());
        return out;
    }
};

// ------------------------------------------------------------------------------------------------------------------ wire formats
// AudioWAV.data(from:sampleRate:normalize:) (Sources/FluidAudio/Shared/AudioConverter.swift:474-532)
struct AudioWAV {
    static std::vector<uint8_t> data(Context &ctx, const std::vector<float> &samples, double sampleRate, bool normalize = true) {
        std::vector<uint8_t> out(static_cast<size_t>(fa_wav_pcm16_size(static_cast<int64_t>(samples.size()))));
        int64_t len = 0;
        ctx.check(fa_wav_encode_pcm16(ctx.handle(), samples.data(), static_cast<int64_t>(samples.size()), sampleRate, normalize ? 1 : 0, out.data(), static_cast<int64_t>(out.size()), &len), "fa_wav_encode_pcm16");
        return out;
    }
};

// RTTMParser.loadSegments (Sources/FluidAudioCLI/Utils/RTTMParser.swift:22-63) on text
struct RTTMParserError : std::runtime_error { using std::runtime_error::runtime_error; };
struct RTTMParser {
    static std::vector<fa_rttm_segment> parse(const std::string &text) {
        int64_t count = 0;
        char bad[512] = {0};
        if (fa_rttm_parse(text.data(), static_cast<int64_t>(text.size()), 1, nullptr, 0, &count, bad, sizeof(bad)) != FA_SUCCESS) throw RTTMParserError(std::string("Invalid RTTM line: ") + bad);
        std::vector<fa_rttm_segment> segs(static_cast<size_t>(count));
        if (count) fa_rttm_parse(text.data(), static_cast<int64_t>(text.size()), 1, segs.data(), count, &count, bad, sizeof(bad));
        return segs;
    }
};

}  // namespace fluidaudio
