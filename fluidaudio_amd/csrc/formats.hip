// formats.hip — the wire formats at the edges of the hot path, so that the engine can be driven and
// checked from files without the Swift host.
//
//   * AudioWAV.data (reference: Sources/FluidAudio/Shared/AudioConverter.swift:474-532): float samples -> optional peak
//     normalisation -> clamp -> 16-bit PCM mono RIFF/WAVE.  The sample pass is a device kernel (max |x| reduction, then
//     x / max -> clamp -> * 32767 -> truncate), bit-identical to the reference's Float arithmetic; the 44-byte header is host code.
//   * RTTM ground-truth lines (Sources/FluidAudioCLI/Utils/RTTMParser.swift:22-63 strict; the benchmark loader
//     Sources/FluidAudioCLI/Commands/SortformerBenchmark.swift:681-731 lenient): "SPEAKER <file> 1 <start> <dur> <NA> <NA> <spk> ...".
//   * the embedding export of OfflineDiarizerManager.exportEmbeddings (:913-955): a JSON array of
//     {chunkIndex, speakerIndex, startFrame, endFrame, startTime, endTime, embedding256, rho128, cluster}.
// A WAV reader and an RTTM writer are provided as extensions (the reference reads audio through AVFoundation and never writes RTTM).
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "fa_common.h"
#include "text_util.h"

namespace {

constexpr int kThreads = 256;

// max |x| of the whole buffer: NaN never wins (Swift's max() over the mapped array keeps the first operand on unordered compares)
__global__ void wav_peak_kernel(const float *__restrict__ x, int64_t n, unsigned int *__restrict__ peak_bits) {
    float m = 0.0f;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float v = fabsf(x[i]);
        if (v > m) m = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off); if (o > m) m = o; }
    if ((threadIdx.x & 63) == 0) atomicMax(peak_bits, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

// s / max (normalised) -> clamp to [-1, 1] -> * 32767 -> Int16 truncation toward zero (:487-495)
__global__ void wav_quantize_kernel(const float *__restrict__ x, int64_t n, const unsigned int *__restrict__ peak_bits, int normalize,
                                    int16_t *__restrict__ out) {
    const float peak = __uint_as_float(*peak_bits);
    const bool scale = normalize && peak > 0.0f;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float s = x[i];
        if (scale) s = __fdiv_rn(s, peak);
        const float c = fmaxf(-1.0f, fminf(1.0f, s));
        out[i] = static_cast<int16_t>(__fmul_rn(c, 32767.0f));
    }
}

void put_u32(uint8_t *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
void put_u16(uint8_t *p, uint16_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }
uint32_t get_u32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | (static_cast<uint32_t>(p[3]) << 24); }
uint16_t get_u16(const uint8_t *p) { return static_cast<uint16_t>(p[0] | (p[1] << 8)); }

using fa_text::parse_float;   // Swift's Float(String): the whole field must be a number (text_util.h)

template <class T>
void append_number(std::string &o, T v) {
    char buf[64];
    if (!(v == v) || v - v != 0) { o += "null"; return; }   // JSONEncoder throws on non-finite values; keep the file parseable
    auto r = std::to_chars(buf, buf + sizeof(buf), v);       // shortest representation that round-trips, like Swift's description
    o.append(buf, r.ptr);
}

}  // namespace

extern "C" {

int64_t fa_wav_pcm16_size(int64_t n_samples) { return n_samples < 0 ? 0 : 44 + 2 * n_samples; }

fa_status fa_wav_encode_pcm16(fa_ctx *ctx, const float *samples, int64_t n, double sample_rate, int32_t normalize, uint8_t *out,
                              int64_t out_capacity, int64_t *out_len) {
    if (!ctx) return FA_INVALID_ARGUMENT;
    if (n < 0 || (n > 0 && !samples) || !out) return fa::set_error(ctx, FA_INVALID_ARGUMENT, "wav: bad arguments");
    const int64_t need = fa_wav_pcm16_size(n);
    if (out_len) *out_len = need;
    if (out_capacity < need) return fa::set_error(ctx, FA_OUTPUT_TOO_SMALL, "wav: output holds %lld of %lld bytes", (long long)out_capacity, (long long)need);
    if (36 + 2 * n > 0xffffffffLL) return fa::set_error(ctx, FA_INDEX_OVERFLOW, "wav: data chunk exceeds 4 GiB");
    memcpy(out, "RIFF", 4); put_u32(out + 4, static_cast<uint32_t>(36 + 2 * n)); memcpy(out + 8, "WAVE", 4);
    memcpy(out + 12, "fmt ", 4); put_u32(out + 16, 16); put_u16(out + 20, 1); put_u16(out + 22, 1);
    put_u32(out + 24, static_cast<uint32_t>(sample_rate)); put_u32(out + 28, static_cast<uint32_t>(sample_rate * 2));
    put_u16(out + 32, 2); put_u16(out + 34, 16);
    memcpy(out + 36, "data", 4); put_u32(out + 40, static_cast<uint32_t>(2 * n));
    if (n == 0) return FA_SUCCESS;
    fa::DeviceGuard guard(ctx->device);
    fa::DevBuf d_x, d_q, d_peak;
    FA_HIP_TRY(ctx, d_x.alloc(sizeof(float) * n));
    FA_HIP_TRY(ctx, d_q.alloc(sizeof(int16_t) * n));
    FA_HIP_TRY(ctx, d_peak.alloc(sizeof(unsigned int)));
    FA_HIP_TRY(ctx, hipMemcpyAsync(d_x.p, samples, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    FA_HIP_TRY(ctx, hipMemsetAsync(d_peak.p, 0, sizeof(unsigned int), ctx->stream));
    const unsigned blocks = static_cast<unsigned>(std::min<int64_t>((n + kThreads - 1) / kThreads, 2048));
    hipLaunchKernelGGL(wav_peak_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_x.as<float>(), n, d_peak.as<unsigned int>());
    hipLaunchKernelGGL(wav_quantize_kernel, dim3(blocks), dim3(kThreads), 0, ctx->stream, d_x.as<float>(), n, d_peak.as<unsigned int>(), normalize,
                       d_q.as<int16_t>());
    FA_HIP_TRY(ctx, hipGetLastError());
    FA_HIP_TRY(ctx, hipMemcpyAsync(out + 44, d_q.p, sizeof(int16_t) * n, hipMemcpyDeviceToHost, ctx->stream));   // little-endian host
    FA_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return FA_SUCCESS;
}

// extension: canonical RIFF/WAVE reader (PCM 16-bit or IEEE float 32-bit, any channel count) -> interleaved float in [-1, 1)
fa_status fa_wav_decode(const uint8_t *data, int64_t len, float *out, int64_t out_capacity, int64_t *frames, int32_t *channels, int32_t *sample_rate) {
    if (!data || len < 12 || memcmp(data, "RIFF", 4) != 0 || memcmp(data + 8, "WAVE", 4) != 0) return FA_INVALID_ARGUMENT;
    int fmt = 0, ch = 0, bits = 0, rate = 0;
    int64_t pos = 12;
    while (pos + 8 <= len) {
        const uint32_t sz = get_u32(data + pos + 4);
        const uint8_t *body = data + pos + 8;
        if (pos + 8 + static_cast<int64_t>(sz) > len && memcmp(data + pos, "data", 4) != 0) return FA_INVALID_ARGUMENT;
        if (memcmp(data + pos, "fmt ", 4) == 0 && sz >= 16) {
            fmt = get_u16(body); ch = get_u16(body + 2); rate = static_cast<int>(get_u32(body + 4)); bits = get_u16(body + 14);
        } else if (memcmp(data + pos, "data", 4) == 0) {
            if (!ch || !((fmt == 1 && bits == 16) || (fmt == 3 && bits == 32))) return FA_INVALID_ARGUMENT;
            const int64_t avail = std::min<int64_t>(sz, len - pos - 8);
            const int64_t n = avail / (bits / 8) / ch * ch;     // whole frames only (a truncated file may end inside one)
            if (frames) *frames = n / ch;
            if (channels) *channels = ch;
            if (sample_rate) *sample_rate = rate;
            if (!out) return FA_SUCCESS;
            if (out_capacity < n) return FA_OUTPUT_TOO_SMALL;
            for (int64_t i = 0; i < n; ++i) {
                if (fmt == 1) out[i] = static_cast<float>(static_cast<int16_t>(get_u16(body + 2 * i))) / 32768.0f;
                else { const uint32_t u = get_u32(body + 4 * i); memcpy(out + i, &u, 4); }
            }
            return FA_SUCCESS;
        }
        pos += 8 + sz + (sz & 1);
    }
    return FA_INVALID_ARGUMENT;
}

// RTTMParser.loadSegments (strict = 1: a malformed line is an error, RTTMParser.swift:37-46) / the benchmark loader (strict = 0:
// malformed lines are skipped and the file order is kept, SortformerBenchmark.swift:700-727).  Returns the number of segments
// through *count even when `out` is too small.
fa_status fa_rttm_parse(const char *text, int64_t len, int32_t strict, fa_rttm_segment *out, int64_t out_capacity, int64_t *count,
                        char *bad_line, int64_t bad_line_capacity) {
    if (!text || len < 0 || !count) return FA_INVALID_ARGUMENT;
    return fa::no_throw(nullptr, "rttm parse", [&]() -> fa_status {
    std::vector<fa_rttm_segment> segs;
    int64_t pos = 0;
    const char *const text_end = text + len;
    while (pos <= len) {
        // lines: components(separatedBy: .newlines) (RTTMParser.swift:30, SortformerBenchmark.swift:699) — \n, \r, \v, \f, U+0085, U+2028, U+2029
        int64_t e = pos;
        int nl = 0;
        while (e < len && (nl = fa_text::nl_len(text + e, text_end)) == 0) ++e;
        const char *a = text + pos, *b = text + e;
        fa_text::trim(a, b, fa_text::ws_len);                  // trimmingCharacters(in: .whitespaces) (:31): Zs + tab
        const std::string line(a, b);
        pos = e + (nl > 0 ? nl : 1);
        if (line.empty() || (strict && line[0] == '#')) continue;
        // fields: split(whereSeparator: \.isWhitespace) (:36) / components(separatedBy: .whitespaces) without the empty ones (:703-705)
        std::vector<std::string> f;
        for (const char *i = line.data(), *le = line.data() + line.size(); i < le;) {
            for (int k; i < le && (k = fa_text::ws_or_nl_len(i, le)) > 0;) i += k;
            const char *j = i;
            while (j < le && fa_text::ws_or_nl_len(j, le) == 0) ++j;
            if (j > i) f.emplace_back(i, j);
            i = j;
        }
        float start = 0, dur = 0;
        const bool ok = f.size() >= 8 && f[0] == "SPEAKER" && parse_float(f[3], start) && parse_float(f[4], dur);
        if (!ok) {
            if (!strict) continue;
            if (bad_line && bad_line_capacity > 0) { const size_t c = std::min<size_t>(line.size(), bad_line_capacity - 1); memcpy(bad_line, line.data(), c); bad_line[c] = 0; }
            *count = static_cast<int64_t>(segs.size());
            return FA_INVALID_ARGUMENT;
        }
        fa_rttm_segment s{};
        s.start_seconds = start;
        s.end_seconds = start + dur;   // Float addition (:49)
        s.quality = 1.0f;
        const size_t c = std::min(f[7].size(), sizeof(s.speaker_id) - 1);
        memcpy(s.speaker_id, f[7].data(), c);
        segs.push_back(s);
    }
    if (strict) std::stable_sort(segs.begin(), segs.end(), [](const fa_rttm_segment &x, const fa_rttm_segment &y) { return x.start_seconds < y.start_seconds; });  // :62
    *count = static_cast<int64_t>(segs.size());
    if (out) {
        if (out_capacity < *count) return FA_OUTPUT_TOO_SMALL;
        std::copy(segs.begin(), segs.end(), out);
    }
    return FA_SUCCESS;
    });
}

// extension: "SPEAKER <file> 1 <start> <duration> <NA> <NA> <speaker> <NA> <NA>\n" per segment, 3 decimals
int64_t fa_rttm_format(const fa_rttm_segment *segs, int64_t n, const char *file_id, char *out, int64_t out_capacity) {
    int64_t length = -1;                                         // -1: the text could not be built (host allocation failed)
    (void)fa::no_throw(nullptr, "rttm format", [&]() -> fa_status {
    std::string o;
    if (!segs) n = 0;
    char buf[256];
    for (int64_t i = 0; i < n; ++i) {
        // the speaker id is a fixed 64-byte field that a caller may have filled to the last byte; a long file id must not truncate the line
        snprintf(buf, sizeof(buf), " 1 %.3f %.3f <NA> <NA> %.64s <NA> <NA>\n", static_cast<double>(segs[i].start_seconds),
                 static_cast<double>(segs[i].end_seconds - segs[i].start_seconds), segs[i].speaker_id);
        o += "SPEAKER ";
        o += file_id ? file_id : "audio";
        o += buf;
    }
    if (out && out_capacity > static_cast<int64_t>(o.size())) memcpy(out, o.c_str(), o.size() + 1);
    length = static_cast<int64_t>(o.size());
    return FA_SUCCESS;
    });
    return length;
}

// exportEmbeddings payload (:918-947) as JSON text.  Returns the length; writes when the buffer is large enough.
int64_t fa_export_embeddings_json(const fa_export_embedding *items, int64_t n, const float *embedding256, int32_t emb_dim, const double *rho128,
                                  int32_t rho_dim, const int32_t *assignments, int64_t n_assignments, char *out, int64_t out_capacity) {
    int64_t length = -1;                                         // -1: the text could not be built (host allocation failed)
    (void)fa::no_throw(nullptr, "embedding export", [&]() -> fa_status {
    std::string o = "[";
    for (int64_t i = 0; i < n; ++i) {
        if (i) o += ',';
        o += "{\"chunkIndex\":"; append_number(o, static_cast<long long>(items[i].chunk_index));
        o += ",\"speakerIndex\":"; append_number(o, static_cast<long long>(items[i].speaker_index));
        o += ",\"startFrame\":"; append_number(o, static_cast<long long>(items[i].start_frame));
        o += ",\"endFrame\":"; append_number(o, static_cast<long long>(items[i].end_frame));
        o += ",\"startTime\":"; append_number(o, items[i].start_time);
        o += ",\"endTime\":"; append_number(o, items[i].end_time);
        o += ",\"embedding256\":[";
        for (int k = 0; k < emb_dim; ++k) { if (k) o += ','; append_number(o, embedding256[i * emb_dim + k]); }
        o += "],\"rho128\":[";
        for (int k = 0; k < rho_dim; ++k) { if (k) o += ','; append_number(o, rho128[i * rho_dim + k]); }
        o += "],\"cluster\":"; append_number(o, static_cast<long long>(i < n_assignments && assignments ? assignments[i] : -1));   // :932-934
        o += '}';
    }
    o += ']';
    if (out && out_capacity > static_cast<int64_t>(o.size())) memcpy(out, o.c_str(), o.size() + 1);
    length = static_cast<int64_t>(o.size());
    return FA_SUCCESS;
    });
    return length;
}

}  // extern "C"
