// asan_text.cpp — the host-only text entries of the C ABI (RTTM reader / writer, ARPA reader + scoring, WAV reader, embedding JSON) under
// AddressSanitizer + UndefinedBehaviorSanitizer: generated and mutated inputs in exact-size heap buffers (no terminator, no slack), so
// that a read past the end of the caller's buffer is a report instead of luck.  Built and run by scripts/asan_text_fuzz.sh from the host
// side of formats.hip / beam.hip (hipcc --cuda-host-only); no GPU, no kernel launch.  Test infrastructure, not part of the product.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "fluidaudio_hip.h"
int main() {
    std::mt19937_64 rng(1);
    auto R = [&](uint64_t n) { return n ? rng() % n : 0; };
    const char *pieces[] = {"SPEAKER", " ", "\t", "\n", "\r", "\xc2\xa0", "\xe2\x80\xa8", "\xe3\x80\x80", "1.5", "-2", "0x1p3", "nan", "<NA>", "spk", "#", "\\data\\", "\\1-grams:", "\\2-grams:", "\\end\\",
                            "ngram 1=2", "the", "cat", "-0.5", "1e400", "\xe2", "\xc2", "\xe2\x80", "", "a very long token ........................................................................ end"};
    const int np = sizeof(pieces) / sizeof(pieces[0]);
    long segs_total = 0, lms = 0;
    for (int it = 0; it < 300000; ++it) {
        std::string t;
        const int nl = static_cast<int>(R(6));
        const char *nums[] = {"1.5", "-2", "0x1p3", "nan", "1e400", ".5", "1_0", "", "inf", "-0.25", "7"};
        const char *seps[] = {" ", "\t", "  ", "\xc2\xa0", "\xe3\x80\x80", "\xe2\x80\x89"};
        const char *ends[] = {"\n", "\r\n", "\r", "\xe2\x80\xa8", "\xc2\x85", "\x0b", ""};
        for (int l = 0; l < nl; ++l) {
            const int kind = static_cast<int>(R(8));
            if (kind < 3) {                                   // RTTM-like line
                const char *sep = seps[R(6)];
                const int nf = 6 + static_cast<int>(R(6));
                for (int f = 0; f < nf; ++f) {
                    if (f) t += sep;
                    if (f == 0) t += R(8) ? "SPEAKER" : "LEXEME";
                    else if (f == 3 || f == 4) t += nums[R(11)];
                    else if (f == 7 && R(4) == 0) t += std::string(static_cast<size_t>(R(200)), 'x');
                    else t += pieces[R(np)];
                }
            } else if (kind < 6) {                            // ARPA-like line
                if (R(4) == 0) t += pieces[15 + R(5)];
                else { t += nums[R(11)]; for (int f = 0, nf = static_cast<int>(R(4)); f < nf; ++f) { t += '\t'; t += R(3) ? pieces[20 + R(3)] : nums[R(11)]; } }
            } else {
                const int k = static_cast<int>(R(12));
                for (int i = 0; i < k; ++i) { t += pieces[R(np)]; if (R(8) == 0) t += static_cast<char>(R(256)); }
            }
            t += ends[R(7)];
        }
        if (R(3) == 0) t = "\\data\\\n\\1-grams:\n" + t;
        if (R(5) == 0) t = "\\2-grams:\n" + t;
        std::vector<char> buf(t.begin(), t.end());                       // exact size, no terminator
        int64_t count = 0;
        char bad[16];
        std::vector<fa_rttm_segment> out(8);
        for (int strict = 0; strict < 2; ++strict) {
            fa_status st = fa_rttm_parse(buf.data(), static_cast<int64_t>(buf.size()), strict, out.data(), 8, &count, bad, sizeof(bad));
            if (st == FA_SUCCESS) {
                segs_total += count;
                std::vector<char> txt(4096);
                fa_rttm_format(out.data(), count, "f", txt.data(), 4096);
            }
        }
        fa_arpa_lm *lm = nullptr;
        if (fa_arpa_parse(nullptr, buf.data(), static_cast<int64_t>(buf.size()), &lm) == FA_SUCCESS && lm) {
            float sc = 0;
            fa_arpa_score(lm, "the", "cat", &sc); fa_arpa_score(lm, "", nullptr, &sc);
            lms += fa_arpa_unigram_count(lm) + fa_arpa_bigram_context_count(lm);
            fa_arpa_destroy(lm);
        }
        // WAV
        std::vector<uint8_t> w(R(120));
        for (auto &b : w) b = static_cast<uint8_t>(R(256));
        if (w.size() >= 12 && R(2)) { memcpy(w.data(), "RIFF", 4); memcpy(w.data() + 8, "WAVE", 4); if (w.size() > 40) { memcpy(w.data() + 12, "fmt ", 4); w[16] = 16; w[17] = w[18] = w[19] = 0; w[20] = R(2) ? 1 : 3; w[21] = 0; w[22] = 1 + R(3); w[23] = 0; w[34] = w[20] == 1 ? 16 : 32; w[35] = 0; memcpy(w.data() + 36, "data", 4); } }
        int64_t frames = 0; int32_t ch = 0, sr = 0;
        if (fa_wav_decode(w.data(), static_cast<int64_t>(w.size()), nullptr, 0, &frames, &ch, &sr) == FA_SUCCESS) {
            std::vector<float> o(static_cast<size_t>(frames * ch));
            fa_wav_decode(w.data(), static_cast<int64_t>(w.size()), o.data(), static_cast<int64_t>(o.size()), nullptr, nullptr, nullptr);
        }
    }
    for (int it = 0; it < 2000; ++it) {
        const int n = static_cast<int>(R(4)), ed = static_cast<int>(R(5)), rd = static_cast<int>(R(5));
        std::vector<fa_export_embedding> items(static_cast<size_t>(n));
        std::vector<float> e(static_cast<size_t>(n * ed));
        std::vector<double> r(static_cast<size_t>(n * rd));
        std::vector<int32_t> as(R(static_cast<uint64_t>(n) + 1));
        for (auto &v : e) v = static_cast<float>(static_cast<int64_t>(R(2000)) - 1000) / 7.0f;
        for (auto &v : r) v = R(10) ? static_cast<double>(R(1000)) / 3.0 : 1e300 * 1e10;
        for (auto &x : items) { x.chunk_index = static_cast<int32_t>(R(100)); x.speaker_index = static_cast<int32_t>(R(3)); x.start_frame = 0; x.end_frame = 9; x.start_time = 0.5; x.end_time = 1.25; }
        const int64_t need = fa_export_embeddings_json(items.data(), n, e.data(), ed, r.data(), rd, as.data(), static_cast<int64_t>(as.size()), nullptr, 0);
        std::vector<char> out(static_cast<size_t>(need) + 1);
        if (fa_export_embeddings_json(items.data(), n, e.data(), ed, r.data(), rd, as.data(), static_cast<int64_t>(as.size()), out.data(), need + 1) != need) { std::printf("json length mismatch\n"); return 1; }
    }
    std::printf("done: %ld segments, %ld lm entries\n", segs_total, lms);
    return 0;
}
