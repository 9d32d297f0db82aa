// A C++ host written against include/fluidaudio.hpp only: the reference's own test cases, restated with the reference's type
// and method names (file:line of the Swift test next to each block).  Prints one line per check; exit code = failed checks.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>

#include "fluidaudio.hpp"

using namespace fluidaudio;

static int failures = 0;
#define CHECK(cond) do { if (cond) std::printf("PASS %s:%d %s\n", __func__, __LINE__, #cond); else { std::printf("FAIL %s:%d %s\n", __func__, __LINE__, #cond); ++failures; } } while (0)

static const char *W = "\xe2\x96\x81";   // U+2581

template <class T> static size_t distinct(const std::vector<T> &v, size_t a, size_t b) { return std::set<T>(v.begin() + a, v.begin() + b).size(); }

static void ahcClusteringTests(Context &ctx) {   // Tests/FluidAudioTests/Diarizer/Offline/AHCClusteringTests.swift:12-145
    AHCClustering ahc{ctx};
    CHECK(ahc.cluster({}, 0.7).empty());
    CHECK(ahc.cluster({{1.0, 0.0, 0.0}}, 0.7) == std::vector<int>{0});
    auto same = ahc.cluster(Matrix(5, {1.0, 2.0, 3.0}), 0.7);
    CHECK(distinct(same, 0, 5) == 1);
    auto g = ahc.cluster({{1, 0, 0}, {.9, .1, 0}, {.95, .05, 0}, {0, 1, 0}, {0, .9, .1}, {0, .95, .05}}, 0.8);
    CHECK(distinct(g, 0, 3) == 1 && distinct(g, 3, 6) == 1 && g[0] != g[3]);
    Matrix e4 = {{1, 0, 0}, {.9, .1, 0}, {0, 1, 0}, {0, .9, .1}};
    CHECK(distinct(ahc.cluster(e4, 0.5), 0, 4) == 2 && distinct(ahc.cluster(e4, 1.5), 0, 4) == 1);
    Matrix eye = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    CHECK(distinct(ahc.cluster(eye, 0.5), 0, 3) == 3 && distinct(ahc.cluster(eye, 2.0), 0, 3) == 1 && distinct(ahc.cluster(eye, 0.0), 0, 3) == 3);
    CHECK(ahc.cluster(Matrix(3), 0.7) == (std::vector<int>{0, 0, 0}));
    CHECK(distinct(ahc.cluster(eye, std::nan("")), 0, 3) == 3);   // NaN threshold -> 0 (:112-116)
}

static void ctcDecoderTests(Context &ctx) {   // Tests/.../CTC/CtcDecoderTests.swift:64-141, 145-260; LogitsArgmaxTests.swift:12-24
    const float L = -100.0f;
    Vocabulary v2 = {{0, std::string(W) + "hello"}, {1, std::string(W) + "world"}}, v1 = {{0, std::string(W) + "hello"}};
    CHECK(ctcGreedyDecode(ctx, {{0, L, L}, {L, L, 0}, {L, 0, L}}, v2, 2) == "hello world");
    CHECK(ctcGreedyDecode(ctx, {{0, L, L}, {0, L, L}, {L, 0, L}}, v2, 2) == "hello world");
    CHECK(ctcGreedyDecode(ctx, {{0, L}, {L, 0}, {0, L}}, v1, 1) == "hello hello");
    CHECK(ctcGreedyDecode(ctx, {{L, 0}, {L, 0}, {L, 0}}, v1, 1) == "");
    CHECK(ctcGreedyDecode(ctx, {}, v1, 1) == "");
    {   // where the [[Float]] overload (CtcDecoder.swift:21-31) and the [1, T, V] overload (:55-64) are defined to differ
        const float nan = std::nanf("");
        Vocabulary abc = {{0, "a"}, {1, "b"}, {2, "c"}};
        CHECK(ctcGreedyDecode(ctx, {{nan, 1.0f, 0.0f}, {0.0f, 0.0f, 3.0f}}, abc, 9) == "ac");          // frame[0] seed: the NaN is never beaten
        const float rect[6] = {nan, 1.0f, 0.0f, 0.0f, 0.0f, 3.0f};
        CHECK(ctcGreedyDecode(ctx, rect, 2, 3, abc, 9) == "bc");                                       // -inf seed: the NaN never wins
        CHECK(ctcGreedyDecode(ctx, {{0.0f, 4.0f}, {}, {0.0f, 4.0f}}, abc, 9) == "b");                  // empty frame skipped before prev
        CHECK(ctcGreedyDecode(ctx, {{0.0f, 1.0f}, {0.0f, 0.0f, 0.0f, 7.0f}, {5.0f}, {9.0f, 1.0f}}, {{0, "a"}, {1, "b"}, {3, "d"}}, 9) == "bda");   // ragged frames
        CHECK(ctcGreedyDecode(ctx, {{nan, nan}, {}, {nan}}, abc, 9) == "a");
    }
    const float m[15] = {0.1f, 0.9f, -0.3f, 0.2f, 0.0f, -2.0f, -1.0f, -0.5f, -3.0f, -4.0f, 7.0f, 7.0f, 8.0f, 8.0f, 1.0f};
    CHECK(argmaxPerFrame(ctx, m, 3, 5, 5) == (std::vector<int>{1, 2, 2}));
    CHECK(ctcBeamSearch(ctx, {{0, L, L}, {L, L, 0}, {L, 0, L}}, v2, nullptr, 5, 0.0f, 0.0f, 2) == "hello world");
    CHECK(ctcBeamSearch(ctx, {{L, 0}, {L, 0}, {L, 0}}, v1, nullptr, 5, 0.0f, 0.0f, 1) == "");
    CHECK(ctcBeamSearch(ctx, {}, v1, nullptr, 5, 0.0f, 0.0f, 1) == "");
    CHECK(ctcBeamSearch(ctx, {{0, L}}, v1, nullptr, 5, 0.0f, 0.0f, 1) == "hello");
}

static void arpaLanguageModelTests(Context &ctx) {   // Tests/.../CTC/ARPALanguageModelTests.swift:39-176
    const std::string arpa = "\\data\\\nngram 1=4\nngram 2=2\n\n\\1-grams:\n-1.0\tthe\t-0.5\n-1.2\tcat\t-0.3\n-1.5\tsat\t0.0\n-2.0\t<unk>\t0.0\n\n"
                             "\\2-grams:\n-0.5\tthe\tcat\n-0.8\tcat\tsat\n\n\\end\\\n";
    ARPALanguageModel lm(arpa);
    const float k = std::log(10.0f);
    auto near = [](float a, float b) { return std::fabs(a - b) < 1e-3f; };
    CHECK(lm.unigramCount() == 4 && lm.bigramContextCount() == 2);
    CHECK(near(lm.score("cat", "the"), -0.5f * k));
    CHECK(near(lm.score("sat", "the"), -0.5f * k - 1.5f * k));
    CHECK(near(lm.score("cat"), -1.2f * k));
    CHECK(near(lm.score("xyzzy"), ARPALanguageModel::unkLogProb));
    CHECK(near(lm.score("xyzzy", "the"), -0.5f * k + ARPALanguageModel::unkLogProb));
    Vocabulary v = {{0, std::string(W) + "the"}, {1, std::string(W) + "cat"}, {2, std::string(W) + "dog"}};
    std::vector<std::vector<float>> lp = {{0.0f, -100.0f, -100.0f, -100.0f}, {-100.0f, -1.0f, -0.9f, -100.0f}};
    CHECK(ctcBeamSearch(ctx, lp, v, nullptr, 10, 0.0f, 0.0f, 3) == "the dog");
    CHECK(ctcBeamSearch(ctx, lp, v, &lm, 10, 5.0f, 0.0f, 3) == "the cat");
}

static void speakerCountAndKMeansTests(Context &ctx) {   // SpeakerCountConstraintsTests.swift:10-136; KMeansClusteringTests.swift:10-131
    auto c = SpeakerCountConstraints::resolve(100, std::nullopt, std::nullopt, std::nullopt);
    CHECK(!c.numSpeakers && c.minSpeakers == 1 && c.maxSpeakers == 100);
    c = SpeakerCountConstraints::resolve(100, 3, 1, 10);
    CHECK(c.numSpeakers == 3 && c.minSpeakers == 3 && c.maxSpeakers == 3);
    c = SpeakerCountConstraints::resolve(5, std::nullopt, 2, 20);
    CHECK(c.minSpeakers == 2 && c.maxSpeakers == 5);
    c = SpeakerCountConstraints::resolve(100, std::nullopt, 10, 5);
    CHECK(c.minSpeakers == 5 && c.maxSpeakers == 5);
    c = SpeakerCountConstraints::resolve(100, -5, std::nullopt, std::nullopt);
    CHECK(c.minSpeakers == 1 && c.maxSpeakers == 1);
    c = SpeakerCountConstraints::resolve(100, std::nullopt, 5, 10);
    CHECK(c.needsAdjustment(3) && c.targetCount(3) == 5 && !c.needsAdjustment(7) && c.targetCount(12) == 10);
    Matrix six = {{1.0, 0.0}, {1.1, 0.1}, {0.0, 1.0}, {0.1, 1.1}, {-1.0, 0.0}, {-0.9, 0.1}};
    auto km = KMeansClustering::clusterWithCentroids(ctx, six, 3, 100, 42);
    CHECK(km.first.size() == 6 && distinct(km.first, 0, 6) == 3 && km.second.size() == 3);
    CHECK(KMeansClustering::clusterWithCentroids(ctx, {{1.0, 0.0}, {1.1, 0.1}, {0.9, 0.2}}, 1, 100, 42).first == (std::vector<int>{0, 0, 0}));
    CHECK(KMeansClustering::clusterWithCentroids(ctx, {{1.0, 0.0}, {0.0, 1.0}}, 5, 100, 42).first == (std::vector<int>{0, 1}));
    CHECK(KMeansClustering::clusterWithCentroids(ctx, six, 3, 300, 12345).first == KMeansClustering::clusterWithCentroids(ctx, six, 3, 300, 12345).first);
    CHECK(distinct(KMeansClustering::clusterWithCentroidsNInit(ctx, six, 3, 100, 10, 0).first, 0, 6) == 3);
}

static void assignmentAndVbxTests(Context &ctx) {   // ConstrainedClusterAssignmentTests.swift:8-63; VBxConstraintTests.swift:93-160
    CHECK(ConstrainedClusterAssignment::assign(ctx, {{0.9, 0.3}, {0.8, 0.6}}, {0, 0}) == (std::vector<int>{0, 1}));
    CHECK(ConstrainedClusterAssignment::assign(ctx, {{0.9, 0.3}, {0.8, 0.6}}, {0, 1}) == (std::vector<int>{0, 0}));
    CHECK(ConstrainedClusterAssignment::assign(ctx, {{0.9}, {0.2}}, {0, 0}) == (std::vector<int>{0, -2}));
    VBxOutput o;
    o.gamma = {{0.7, 0.1, 0.1, 0.05, 0.05}, {0.1, 0.7, 0.1, 0.05, 0.05}, {0.1, 0.1, 0.7, 0.05, 0.05}, {0.6, 0.2, 0.1, 0.05, 0.05}, {0.2, 0.6, 0.1, 0.05, 0.05}, {0.1, 0.2, 0.6, 0.05, 0.05}};
    o.pi = {0.4, 0.3, 0.28, 0.01, 0.01};
    o.numClusters = 5;
    CHECK(o.assignedClusterCount() == 3 && SpeakerCountConstraints::resolve(6, 5, std::nullopt, std::nullopt).needsAdjustment(o.assignedClusterCount()));
    // two well separated speakers in a synthetic PLDA space: refine keeps them, numSpeakers = 3 forces the K-Means fallback
    Matrix rho, emb;
    std::vector<int> init;
    for (int i = 0; i < 40; ++i) {
        const double s = i % 2 ? 1.0 : -1.0, j = 0.01 * (i % 7);
        std::vector<double> row(8);
        for (int k = 0; k < 8; ++k) row[k] = 8.0 * s * (((k * 7) % 5) - 2) / 2.0 + 0.01 * ((i * 31 + k * 17) % 13 - 6);
        rho.push_back(row);
        emb.push_back({s + j, 1.0 - s, 0.5 * j});
        init.push_back(i % 2);
    }
    VBxClustering vbx{ctx, {2.0, 13.0 / 7.0, 12.0 / 7.0, 11.0 / 7.0, 10.0 / 7.0, 9.0 / 7.0, 8.0 / 7.0, 1.0}};
    VBxOutput r = vbx.refine(rho, init);
    CHECK(r.numClusters == 2 && r.assignedClusterCount() == 2 && !r.wasAdjusted && r.hardClusters[0] != r.hardClusters[1]);
    VBxOutput f = vbx.refineWithConstraints(rho, emb, init, SpeakerCountConstraints::resolve(40, 3, std::nullopt, std::nullopt));
    CHECK(f.wasAdjusted && f.numClusters == 3 && f.originalClusterCount == 2 && distinct(f.hardClusters, 0, 40) == 3 && f.centroids.size() == 3);
}

static void composedStageTests(Context &ctx) {   // OfflineDiarizerManager.cluster (:270-375) through the single call; LuxTtsMelExtractorTests.swift:18-41 (shapes)
    // 3 speakers, 30 two-second windows with 3 local slots each: unit centres + small deterministic jitter
    std::vector<std::vector<float>> emb;
    Matrix rho;
    std::vector<int> chunks, truth;
    const double centre[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    for (int w = 0; w < 30; ++w)
        for (int s = 0; s < 3; ++s) {
            const int spk = (s + w) % 3;
            std::vector<float> e(8, 0.0f);
            std::vector<double> r(6, 0.0);
            for (int k = 0; k < 4; ++k) e[k] = static_cast<float>(centre[spk][k] + 0.01 * (((w * 7 + s * 3 + k) % 5) - 2));
            for (int k = 0; k < 6; ++k) r[k] = 6.0 * ((spk == k % 3) ? 1.0 : -0.5) + 0.05 * (((w * 5 + k) % 7) - 3);
            emb.push_back(e); rho.push_back(r); chunks.push_back(w); truth.push_back(spk);
        }
    const std::vector<double> phi{2.0, 1.8, 1.6, 1.4, 1.2, 1.0};
    auto res = clusterEmbeddings(ctx, emb, rho, chunks, phi);
    bool pure = res.centroids.size() == 3 && res.info.constrained == 1 && res.info.training_rows == 90;
    for (size_t i = 0; pure && i < truth.size(); ++i) for (size_t j = 0; pure && j < truth.size(); ++j) pure = (truth[i] == truth[j]) == (res.assignments[i] == res.assignments[j]);
    CHECK(pure);
    // the same through the single-stage mirrors (AHCClustering -> VBxClustering -> ...): identical initial partition size
    Matrix e64;
    for (const auto &e : emb) e64.emplace_back(e.begin(), e.end());
    CHECK(static_cast<int>(distinct(AHCClustering{ctx}.cluster(e64, 0.6), 0, 90)) == res.info.initial_clusters);
    OfflineClusteringConfig forced;
    forced.numSpeakers = 2;
    auto f2 = clusterEmbeddings(ctx, emb, rho, chunks, phi, forced);
    CHECK(f2.info.was_adjusted == 1 && f2.info.constrained == 0 && f2.centroids.size() == 2 && distinct(f2.assignments, 0, 90) == 2);
    bool threw = false;
    try { clusterEmbeddings(ctx, {}, {}, {}, phi); } catch (const Error &) { threw = true; }
    CHECK(threw);
    // the same recording twice plus an empty one through the batched call: per recording the single call's result, the empty one fails alone
    std::vector<OfflineRecording> recs{{emb, rho, chunks}, {{}, {}, {}}, {emb, rho, chunks}};
    auto many = clusterEmbeddingsBatch(ctx, recs, phi);
    CHECK(many.size() == 3 && many[0] && !many[1] && many[2] && many[0]->assignments == res.assignments && many[2]->assignments == res.assignments &&
          many[0]->centroids == res.centroids);
    LuxTtsMelExtractor lux{ctx};
    std::vector<float> a(24000);
    for (size_t i = 0; i < a.size(); ++i) a[i] = 0.2f * std::sin(0.03f * static_cast<float>(i));
    auto m = lux.extract(a);
    CHECK(lux.frameCount(24000) == 94 && m.size() == 94 && m[0].size() == 100 && lux.extract({}).empty());
    bool finite = true;
    for (const auto &row : m) for (float v : row) finite = finite && std::isfinite(v) && v >= std::log(1e-7f) - 1e-3f;
    CHECK(finite);
}

static void melAndFormatTests(Context &ctx) {   // AudioMelSpectrogramTests.swift:22-122 (shapes, frame counts); AudioConverter.swift:474-532
    AudioMelSpectrogram mel(ctx);
    std::vector<float> a(16000);
    for (size_t i = 0; i < a.size(); ++i) a[i] = 0.1f * std::sin(0.05f * static_cast<float>(i));
    auto f = mel.computeFlat(a);
    CHECK(f.melLength == 101 && f.numFrames == 101 && f.mel.size() == 128u * 101u);
    auto t = mel.computeFlatTransposed(a);
    bool transposed = t.melLength == 101 && t.mel.size() == f.mel.size();
    for (int m = 0; transposed && m < 128; m += 17) for (int k = 0; k < 101; k += 9) transposed = f.mel[static_cast<size_t>(m) * 101 + k] == t.mel[static_cast<size_t>(k) * 128 + m];
    CHECK(transposed);
    auto e = mel.computeFlat({});
    CHECK(e.melLength == 0 && e.numFrames == 1 && e.mel == std::vector<float>(128, 0.0f));   // :199-201
    CHECK(mel.hannWindow().size() == 400 && mel.hannWindow()[0] == 0.0f && mel.melFilterbankFlat().size() == 128u * 257u);
    AudioMelSpectrogram padded(ctx, 16000, 128, 512, 160, 400, 0.97f, 16);
    CHECK(padded.computeFlat(a).numFrames == 112 && padded.computeFlat(a).melLength == 101);
    auto wav = AudioWAV::data(ctx, {0.5f, -0.25f}, 16000.0);
    CHECK(wav.size() == 48 && std::memcmp(wav.data(), "RIFF", 4) == 0 && wav[44] == 0xff && wav[45] == 0x7f && wav[46] == 0x01 && wav[47] == 0xc0);   // 32767, -16383
    auto segs = RTTMParser::parse("SPEAKER m 1 2.0 1.0 <NA> <NA> B <NA> <NA>\nSPEAKER m 1 0.5 1.0 <NA> <NA> A <NA> <NA>\n");
    CHECK(segs.size() == 2 && std::string(segs[0].speaker_id) == "A" && segs[1].end_seconds == 3.0f);
    bool threw = false;
    try { RTTMParser::parse("SPEAKER m 1 x 1.0 <NA> <NA> B\n"); } catch (const RTTMParserError &) { threw = true; }
    CHECK(threw);
}

static void devicePoolTests() {   // several OfflineDiarizerManager-style workers in one process (OfflineDiarizerManager.swift:270), one lease each
    DevicePool pool(std::vector<int32_t>{0, 0});
    CHECK(pool.size() == 2);
    {
        auto a = pool.lease();
        auto b = pool.lease();
        CHECK(a.ctx != b.ctx && a.device() == 0 && b.device() == 0);
    }
    std::vector<std::vector<double>> data(3), z;
    for (int p = 0; p < 3; ++p)
        for (int i = 0; i < 40 + 13 * p; ++i) for (int k = 0; k < 5; ++k) data[static_cast<size_t>(p)].push_back(std::sin(0.37 * (i * 5 + k) + p) + 0.01 * i);
    auto st = pool.linkageMany(data, 5, z);
    bool ok = st.size() == 3;
    for (int p = 0; ok && p < 3; ++p) {
        const size_t n = data[static_cast<size_t>(p)].size() / 5;
        std::vector<double> alone((n - 1) * 4);
        ok = st[static_cast<size_t>(p)] == FA_SUCCESS &&
             fastcluster_compute_centroid_linkage(data[static_cast<size_t>(p)].data(), n, 5, alone.data(), alone.size()) == FASTCLUSTER_WRAPPER_SUCCESS &&
             alone == z[static_cast<size_t>(p)];
    }
    CHECK(ok);
}

int main(int argc, char **argv) {
    if (argc > 1 && std::strcmp(argv[1], "link") == 0) {   // CPU tier: the header compiles, the library links, host-only pieces work
        ARPALanguageModel lm("\\1-grams:\n-1.0\ta\n\\end\\\n");
        CHECK(lm.unigramCount() == 1);
        CHECK(SpeakerCountConstraints::resolve(100, 0, std::nullopt, std::nullopt).maxSpeakers == 1);
        CHECK(decodeCtcTokenIds({0, 1}, {{0, std::string(W) + "a"}, {1, std::string(W) + "b"}}) == "a b");
        // OfflineModuleTests.swift:10-33: the default configuration validates, threshold 2.5 is an invalidConfiguration naming clustering.threshold
        OfflineClusteringConfig().validate();
        OfflineClusteringConfig bad;
        bad.clusteringThreshold = 2.5;
        bool invalid = false;
        try { bad.validate(); } catch (const Error &e) { invalid = e.status == FA_INVALID_ARGUMENT && std::string(e.what()).find("clustering.threshold") != std::string::npos; }
        CHECK(invalid);
        bool threw = false;
        try { Context c(0); } catch (const Error &e) { threw = e.status == FA_RUNTIME_ERROR; }
        std::printf("context without a GPU throws RUNTIME_ERROR: %s\n", threw ? "yes" : "no (a GPU is present)");
        return failures;
    }
    Context ctx(0);
    ahcClusteringTests(ctx);
    ctcDecoderTests(ctx);
    arpaLanguageModelTests(ctx);
    speakerCountAndKMeansTests(ctx);
    assignmentAndVbxTests(ctx);
    melAndFormatTests(ctx);
    composedStageTests(ctx);
    devicePoolTests();
    std::printf("%d failed\n", failures);
    return failures;
}
