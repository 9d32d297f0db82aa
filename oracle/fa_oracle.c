/*
 * fa_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See fa_oracle.h.
 *
 * Build with -O2 -ffp-contract=off: the restatement keeps one rounding per
 * floating-point operation, like the reference's scalar Swift / non-FMA C++ build.
 */
#define _GNU_SOURCE
#include "fa_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ================================ mel ========================================= */

void fa_oracle_mel_default_config(fa_oracle_mel_config *c) {
    /* AudioMelSpectrogram.swift:59-70 defaults */
    c->sample_rate = 16000;
    c->n_mels = 128;
    c->n_fft = 512;
    c->hop = 160;
    c->win = 400;
    c->preemph = 0.97f;
    c->pad_to = 0;
    c->log_floor = ldexpf(1.0f, -24);
    c->floor_clamped = 0;
    c->window_periodic = 0;
}

/* AudioMelSpectrogram.swift:553-562 */
void fa_oracle_hann(int win, int periodic, float *out) {
    const float divisor = periodic ? (float)win : (float)(win - 1);
    const float pi_f = (float)M_PI; /* Float.pi */
    for (int i = 0; i < win; ++i) {
        const float phase = 2.0f * pi_f * (float)i / divisor;
        out[i] = 0.5f * (1.0f - cosf(phase));
    }
}

/* AudioMelSpectrogram.swift:575-599 */
static float hz_to_mel(float hz) {
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float log_step = logf(6.4f) / 27.0f;
    if (hz >= min_log_hz) return min_log_mel + logf(hz / min_log_hz) / log_step;
    return hz / f_sp;
}
static float mel_to_hz(float mel) {
    const float f_sp = 200.0f / 3.0f;
    const float min_log_hz = 1000.0f;
    const float min_log_mel = min_log_hz / f_sp;
    const float log_step = logf(6.4f) / 27.0f;
    if (mel >= min_log_mel) return min_log_hz * expf(log_step * (mel - min_log_mel));
    return f_sp * mel;
}

/* AudioMelSpectrogram.swift:564-642 */
void fa_oracle_slaney_filterbank(int n_fft, int n_mels, int sample_rate, float *out) {
    const int bins = n_fft / 2 + 1;
    const float f_min = 0.0f, f_max = (float)sample_rate / 2.0f;
    const float mel_min = hz_to_mel(f_min), mel_max = hz_to_mel(f_max);
    float *pts = (float *)malloc(sizeof(float) * (size_t)(n_mels + 2));
    float *freqs = (float *)malloc(sizeof(float) * (size_t)bins);
    for (int i = 0; i < n_mels + 2; ++i) {
        const float mel = mel_min + (float)i * (mel_max - mel_min) / (float)(n_mels + 1);
        pts[i] = mel_to_hz(mel);
    }
    for (int i = 0; i < bins; ++i) freqs[i] = (float)i * (float)sample_rate / (float)n_fft;
    memset(out, 0, sizeof(float) * (size_t)n_mels * (size_t)bins);
    for (int m = 0; m < n_mels; ++m) {
        const float fl = pts[m], fc = pts[m + 1], fr = pts[m + 2];
        const float norm = 2.0f / (fr - fl);
        for (int k = 0; k < bins; ++k) {
            const float f = freqs[k];
            if (f >= fl && f < fc) out[(size_t)m * bins + k] = norm * (f - fl) / (fc - fl);
            else if (f >= fc && f <= fr) out[(size_t)m * bins + k] = norm * (fr - f) / (fr - fc);
        }
    }
    free(pts);
    free(freqs);
}

/* fp32 iterative radix-2 DIT FFT, twiddles rounded from double.  Stand-in for
 * vDSP_DFT_zop (AudioMelSpectrogram.swift:107-111,471) whose internal order is closed. */
void fa_oracle_fft_f32(int n, float *re, float *im) {
    int j = 0;
    for (int i = 1; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            float t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1;
        for (int k = 0; k < half; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)len;
            const float wr = (float)cos(ang), wi = (float)sin(ang);
            for (int s = k; s < n; s += len) {
                const int t = s + half;
                const float xr = re[t] * wr - im[t] * wi;
                const float xi = re[t] * wi + im[t] * wr;
                re[t] = re[s] - xr; im[t] = im[s] - xi;
                re[s] = re[s] + xr; im[s] = im[s] + xi;
            }
        }
    }
}

int fa_oracle_mel_frames_center(const fa_oracle_mel_config *c, long n) {
    /* :195-197, :341-343 */
    const long padded = n + 2 * (long)(c->n_fft / 2);
    const long frames = 1 + (padded - c->win) / c->hop;
    if (frames <= 0 || n <= 0) return 0;
    return (int)frames;
}
int fa_oracle_mel_frames_prepadded(const fa_oracle_mel_config *c, long n) {
    /* :345 */
    long frames = (n - c->n_fft) / c->hop + 1;
    if (frames < 0) frames = 0;
    if (n <= 0) return 0;
    return (int)frames;
}
int fa_oracle_mel_padded_frames(const fa_oracle_mel_config *c, int frames) {
    const int pad_to = c->pad_to > 1 ? c->pad_to : 1; /* :72 */
    return ((frames + pad_to - 1) / pad_to) * pad_to; /* :204 == :354 for frames>0 */
}

static float log_value(const fa_oracle_mel_config *c, float v) {
    /* :542-549 */
    if (c->floor_clamped) return logf(v > c->log_floor ? v : c->log_floor);
    return logf(v + c->log_floor);
}

/* shared body of computeFlat / computeFlatTransposed (:185-292, :325-456) */
static int mel_body(const fa_oracle_mel_config *c, const float *audio, long n, float last,
                    int prepadded, int frames, int transposed, int special_case_zero_preemph,
                    float *out, int *mel_length, int *num_frames) {
    const int nfft = c->n_fft, bins = nfft / 2 + 1, nm = c->n_mels;
    if (frames <= 0 || n <= 0) {
        /* guard (:199-201, :349-351): [padValue]*nMels, melLength 0, numFrames 1 */
        for (int m = 0; m < nm; ++m) out[m] = 0.0f;
        *mel_length = 0;
        *num_frames = 1;
        return 0;
    }
    const int tpad = fa_oracle_mel_padded_frames(c, frames);
    const long pad = prepadded ? 0 : nfft / 2;
    const long pc = n + 2 * pad;
    float *padded = (float *)calloc((size_t)pc, sizeof(float));
    float *hann = (float *)malloc(sizeof(float) * (size_t)c->win);
    float *fb = (float *)malloc(sizeof(float) * (size_t)nm * (size_t)bins);
    float *re = (float *)malloc(sizeof(float) * (size_t)nfft);
    float *im = (float *)malloc(sizeof(float) * (size_t)nfft);
    float *pw = (float *)malloc(sizeof(float) * (size_t)bins);
    if (!padded || !hann || !fb || !re || !im || !pw) return -1;
    fa_oracle_hann(c->win, c->window_periodic, hann);
    fa_oracle_slaney_filterbank(nfft, nm, c->sample_rate, fb);

    if (special_case_zero_preemph && c->preemph == 0.0f) {
        memcpy(padded + pad, audio, sizeof(float) * (size_t)n); /* :363-371 */
    } else {
        padded[pad] = audio[0] - c->preemph * last; /* :211, :373 */
        const float neg = -c->preemph;
        for (long i = 0; i + 1 < n; ++i) padded[pad + 1 + i] = audio[i] * neg + audio[i + 1]; /* vDSP_vsma :219-225 */
    }
    memset(out, 0, sizeof(float) * (size_t)nm * (size_t)tpad);
    const int off = (nfft - c->win) / 2; /* :234 */
    for (int t = 0; t < frames; ++t) {
        memset(re, 0, sizeof(float) * (size_t)nfft);
        memset(im, 0, sizeof(float) * (size_t)nfft);
        const long a = (long)t * c->hop + off;
        long avail = pc - a;
        if (avail > c->win) avail = c->win; /* :248 */
        for (long i = 0; i < avail; ++i) re[off + i] = padded[a + i] * hann[i];
        fa_oracle_fft_f32(nfft, re, im);
        for (int k = 0; k < bins; ++k) {
            const float r2 = re[k] * re[k], i2 = im[k] * im[k]; /* vsq, vsq, vadd :476-480 */
            pw[k] = r2 + i2;
        }
        for (int m = 0; m < nm; ++m) {
            float s = 0.0f;
            const float *row = fb + (size_t)m * bins;
            for (int k = 0; k < bins; ++k) s += row[k] * pw[k]; /* vDSP_mmul :273-281 */
            const float v = log_value(c, s);
            if (transposed) out[(size_t)t * nm + m] = v; /* :451 */
            else out[(size_t)m * tpad + t] = v;          /* :287 */
        }
    }
    free(padded); free(hann); free(fb); free(re); free(im); free(pw);
    *mel_length = frames;
    *num_frames = tpad;
    return 0;
}

int fa_oracle_mel_flat(const fa_oracle_mel_config *c, const float *audio, long n, float last,
                       float *out, int *mel_length, int *num_frames) {
    /* computeFlat has no preemph==0 special case (:206-227) */
    return mel_body(c, audio, n, last, 0, fa_oracle_mel_frames_center(c, n), 0, 0, out, mel_length, num_frames);
}

int fa_oracle_mel_flat_transposed(const fa_oracle_mel_config *c, const float *audio, long n, float last,
                                  int prepadded, int expected_frames, float *out, int *mel_length,
                                  int *num_frames) {
    int frames = prepadded ? fa_oracle_mel_frames_prepadded(c, n) : fa_oracle_mel_frames_center(c, n);
    if (expected_frames >= 0 && n > 0) frames = expected_frames; /* :347 */
    return mel_body(c, audio, n, last, prepadded, frames, 1, 1, out, mel_length, num_frames);
}

int fa_oracle_mel_legacy(const fa_oracle_mel_config *c, const float *audio, long n, float *out) {
    /* compute (:132-178) */
    if (n < c->win) {
        /* Swift Int division truncates toward zero: 1 + (n-win)/hop can still be 1 for
         * -hop < n-win < 0; the loop then windows what exists (:148-153). */
    }
    const long frames = 1 + (n - c->win) / c->hop;
    if (frames <= 0) return 0;
    const int nfft = c->n_fft, bins = nfft / 2 + 1, nm = c->n_mels;
    float *hann = (float *)malloc(sizeof(float) * (size_t)c->win);
    float *fb = (float *)malloc(sizeof(float) * (size_t)nm * (size_t)bins);
    float *re = (float *)malloc(sizeof(float) * (size_t)nfft);
    float *im = (float *)malloc(sizeof(float) * (size_t)nfft);
    fa_oracle_hann(c->win, c->window_periodic, hann);
    fa_oracle_slaney_filterbank(nfft, nm, c->sample_rate, fb);
    for (long t = 0; t < frames; ++t) {
        memset(re, 0, sizeof(float) * (size_t)nfft);
        memset(im, 0, sizeof(float) * (size_t)nfft);
        for (int i = 0; i < c->win; ++i) {
            const long idx = t * c->hop + i;
            if (idx >= 0 && idx < n) re[i] = audio[idx] * hann[i];
        }
        fa_oracle_fft_f32(nfft, re, im);
        for (int m = 0; m < nm; ++m) {
            float s = 0.0f;
            for (int k = 0; k < bins; ++k) {
                const float p = re[k] * re[k] + im[k] * im[k]; /* :522 */
                s += fb[(size_t)m * bins + k] * p;             /* :534 */
            }
            out[(size_t)m * frames + t] = log_value(c, s);
        }
    }
    free(hann); free(fb); free(re); free(im);
    return (int)frames;
}

int fa_oracle_logmel_generic(const float *audio, long n, int n_fft, int hop, const float *window,
                             const float *fb, int n_mels, int power, float floor_v, int frames,
                             float *out) {
    /* TTS/LuxTts/LuxTtsMelExtractor.swift:52-132 */
    if (n <= 0 || frames <= 0) return 0;
    const int pad = n_fft / 2, bins = n_fft / 2 + 1;
    float *padded = (float *)calloc((size_t)(n + 2 * pad), sizeof(float));
    float *re = (float *)malloc(sizeof(float) * (size_t)n_fft);
    float *im = (float *)malloc(sizeof(float) * (size_t)n_fft);
    float *mag = (float *)malloc(sizeof(float) * (size_t)bins);
    for (int i = 0; i < pad; ++i) { /* :63-66 */
        long a = pad - i; if (a > n - 1) a = n - 1;
        long b = n - 2 - i; if (b < 0) b = 0;
        padded[i] = audio[a];
        padded[pad + n + i] = audio[b];
    }
    memcpy(padded + pad, audio, sizeof(float) * (size_t)n);
    const long stft_frames = 1 + n / hop; /* :70 */
    int produced = 0;
    for (int t = 0; t < frames && t < stft_frames; ++t) {
        const long start = (long)t * hop;
        for (int i = 0; i < n_fft; ++i) { re[i] = padded[start + i] * window[i]; im[i] = 0.0f; }
        fa_oracle_fft_f32(n_fft, re, im);
        for (int k = 0; k < bins; ++k) {
            const float r2 = re[k] * re[k], i2 = im[k] * im[k];
            const float p = r2 + i2;
            mag[k] = power == 1 ? sqrtf(p) : p;
        }
        for (int m = 0; m < n_mels; ++m) {
            float s = 0.0f;
            for (int k = 0; k < bins; ++k) s += fb[(size_t)m * bins + k] * mag[k];
            out[(size_t)t * n_mels + m] = logf(s > floor_v ? s : floor_v);
        }
        produced = t + 1;
    }
    for (int t = produced; t < frames && produced > 0; ++t) /* :124-126 replicate last frame */
        memcpy(out + (size_t)t * n_mels, out + (size_t)(produced - 1) * n_mels, sizeof(float) * (size_t)n_mels);
    free(padded); free(re); free(im); free(mag);
    return frames;
}

/* ============================ argmax + CTC greedy ================================= */

static float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 0x3ffu) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 127 - 15) << 23 | man << 13;
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

void fa_oracle_argmax_rows(const void *logits, int is_f16, long frames, long vocab, long row_stride,
                           int32_t *ids) {
    /* LogitsArgmax.swift:22-28 (+ :33-52 for fp16); tie/NaN rule of CtcDecoder.swift:55-64:
     * seed -inf at index 0, strict '>' => lowest index wins ties, NaN never wins. */
    for (long t = 0; t < frames; ++t) {
        float best = -INFINITY;
        int32_t bi = 0;
        for (long v = 0; v < vocab; ++v) {
            const float x = is_f16 ? half_to_float(((const uint16_t *)logits)[t * row_stride + v])
                                   : ((const float *)logits)[t * row_stride + v];
            if (x > best) { best = x; bi = (int32_t)v; }
        }
        ids[t] = bi;
    }
}

long fa_oracle_ctc_collapse(const int32_t *frame_ids, long frames, int32_t blank_id, int32_t *out) {
    /* CtcDecoder.swift:52-68 ; SenseVoiceManager.swift:119-126 */
    long n = 0;
    int32_t prev = -1;
    for (long t = 0; t < frames; ++t) {
        const int32_t b = frame_ids[t];
        if (b != blank_id && b != prev) out[n++] = b;
        prev = b;
    }
    return n;
}

long fa_oracle_ctc_greedy(const void *logits, int is_f16, long frames, long vocab, long row_stride,
                          int32_t blank_id, int32_t *out) {
    if (frames <= 0) return 0;
    int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)frames);
    fa_oracle_argmax_rows(logits, is_f16, frames, vocab, row_stride, ids);
    const long n = fa_oracle_ctc_collapse(ids, frames, blank_id, out);
    free(ids);
    return n;
}

long fa_oracle_ctc_greedy_rows(const float *values, const int64_t *row_offsets, long rows, int32_t blank_id,
                               int32_t *frame_ids, int32_t *out) {
    /* CtcDecoder.swift:15-36 — the [[Float]] overload, statement for statement: empty frames are skipped BEFORE prev is
     * touched (:23), the scan is seeded with frame[0] (:24-25) and runs over 1..<frame.count (:26) with a strict '>' (:27);
     * a NaN in frame[0] therefore wins (nothing compares greater than NaN), unlike the -inf seed of the [1,T,V] overload
     * (:55-64, fa_oracle_ctc_greedy above).  frame_ids (optional): the per-frame winner, -1 for an empty frame. */
    long n = 0;
    int32_t prev = -1;
    for (long t = 0; t < rows; ++t) {
        const float *frame = values + row_offsets[t];
        const int64_t count = row_offsets[t + 1] - row_offsets[t];
        if (count <= 0) { if (frame_ids) frame_ids[t] = -1; continue; }
        int32_t bestIdx = 0;
        float bestVal = frame[0];
        for (int64_t v = 1; v < count; ++v) {
            if (frame[v] > bestVal) { bestVal = frame[v]; bestIdx = (int32_t)v; }
        }
        if (frame_ids) frame_ids[t] = bestIdx;
        if (bestIdx != blank_id && bestIdx != prev) out[n++] = bestIdx;
        prev = bestIdx;
    }
    return n;
}

/* ================================ AHC pre/post ==================================== */

void fa_oracle_ahc_normalize(const double *x, long n, long d, double *out) {
    /* AHCClustering.swift:70-105 */
    for (long i = 0; i < n; ++i) {
        double norm = 0.0;
        for (long k = 0; k < d; ++k) norm += x[i * d + k] * x[i * d + k]; /* vDSP_dotprD */
        const double scale = norm > 0 ? 1.0 / sqrt(norm) : 0.0;
        for (long k = 0; k < d; ++k) out[i * d + k] = x[i * d + k] * scale; /* vDSP_vsmulD */
    }
}

double fa_oracle_ahc_clamp_threshold(double thr) {
    /* AHCClustering.swift:112-121 */
    if (thr != thr) return 0.0;
    if (thr < 0.0) return 0.0;
    if (thr > 2.0) return 2.0;
    return thr;
}

void fa_oracle_ahc_cut(const double *z, long n, double thr, int32_t *labels) {
    /* AHCClustering.swift:124-197 then :200-210 */
    if (n <= 0) return;
    if (n == 1) { labels[0] = 0; return; }
    const long total = 2 * n - 1;
    long *left = (long *)malloc(sizeof(long) * (size_t)total);
    long *right = (long *)malloc(sizeof(long) * (size_t)total);
    double *h = (double *)calloc((size_t)total, sizeof(double));
    long *stack = (long *)malloc(sizeof(long) * (size_t)(2 * total + 2));
    long *queue = (long *)malloc(sizeof(long) * (size_t)(2 * total + 2));
    long *assign = (long *)malloc(sizeof(long) * (size_t)n);
    for (long i = 0; i < total; ++i) left[i] = right[i] = -1;
    for (long r = 0; r < n - 1; ++r) {
        left[n + r] = (long)z[4 * r];
        right[n + r] = (long)z[4 * r + 1];
        h[n + r] = z[4 * r + 2];
    }
    for (long i = 0; i < n; ++i) assign[i] = -1;
    long sp = 0, next = 0;
    stack[sp++] = total - 1;
    while (sp > 0) {
        const long node = stack[--sp];
        if (node < 0) continue;
        if (node < n) {
            if (assign[node] == -1) assign[node] = next++;
            continue;
        }
        if (h[node] <= thr) {
            const long label = next++;
            long qp = 0;
            queue[qp++] = node;
            while (qp > 0) {
                const long cur = queue[--qp];
                if (cur < n) assign[cur] = label;
                else {
                    if (left[cur] >= 0) queue[qp++] = left[cur];
                    if (right[cur] >= 0) queue[qp++] = right[cur];
                }
            }
        } else {
            if (left[node] >= 0) stack[sp++] = left[node];
            if (right[node] >= 0) stack[sp++] = right[node];
        }
    }
    for (long i = 0; i < n; ++i) if (assign[i] == -1) assign[i] = next++;
    /* remapClusterIds: first-appearance order */
    long *map = (long *)malloc(sizeof(long) * (size_t)(next + 1));
    for (long i = 0; i <= next; ++i) map[i] = -1;
    long nid = 0;
    for (long i = 0; i < n; ++i) {
        if (map[assign[i]] < 0) map[assign[i]] = nid++;
        labels[i] = (int32_t)map[assign[i]];
    }
    free(left); free(right); free(h); free(stack); free(queue); free(assign); free(map);
}

int fa_oracle_ahc_cluster(fa_oracle_linkage_fn linkage, const double *x, long n, long d,
                          double threshold, int32_t *labels) {
    /* AHCClustering.swift:20-67 */
    if (n <= 0) return 0;
    if (d <= 0) { for (long i = 0; i < n; ++i) labels[i] = 0; return 0; }
    if (n == 1) { labels[0] = 0; return 0; }
    double *norm = (double *)malloc(sizeof(double) * (size_t)n * (size_t)d);
    double *z = (double *)calloc((size_t)(n - 1) * 4, sizeof(double));
    fa_oracle_ahc_normalize(x, n, d, norm);
    const int status = linkage(norm, (size_t)n, (size_t)d, z, (size_t)(n - 1) * 4);
    if (status != 0) {
        for (long i = 0; i < n; ++i) labels[i] = (int32_t)i; /* :52-55 */
    } else {
        fa_oracle_ahc_cut(z, n, fa_oracle_ahc_clamp_threshold(threshold), labels);
    }
    free(norm); free(z);
    return status;
}

static double sqeuclid(const double *a, const double *b, size_t d) {
    /* FastClusterWrapper.cpp:45-52,68-75 */
    double s = 0.0;
    for (size_t k = 0; k < d; ++k) { const double diff = a[k] - b[k]; s += diff * diff; }
    return s;
}

int fa_oracle_linkage_naive(const double *data, size_t n, size_t d, double *z, size_t zlen) {
    /* status contract: FastClusterWrapper.cpp:203-243 */
    if (!data || !z) return 1;
    if (n == 0) return 0;
    if (d == 0) return 1;
    if (n > 0x7fffffffu || d > 0x7fffffffu) return 2;
    if (zlen < (n > 1 ? (n - 1) * 4 : 0)) return 3;
    if (n == 1) return 0;
    const size_t total = 2 * n - 1;
    double *pts = (double *)malloc(sizeof(double) * total * d);
    double *size = (double *)malloc(sizeof(double) * total);
    char *active = (char *)calloc(total, 1);
    if (!pts || !size || !active) { free(pts); free(size); free(active); return 4; }
    memcpy(pts, data, sizeof(double) * n * d);
    for (size_t i = 0; i < n; ++i) { active[i] = 1; size[i] = 1.0; }
    int status = 0;
    for (size_t step = 0; step + 1 < n && status == 0; ++step) {
        double best = INFINITY;
        size_t bi = 0, bj = 0;
        int found = 0;
        for (size_t i = 0; i < n + step; ++i) {
            if (!active[i]) continue;
            for (size_t j = 0; j < i; ++j) {
                if (!active[j]) continue;
                const double v = sqeuclid(pts + i * d, pts + j * d, d);
                if (v != v) { status = 5; break; }
                if (!found || v < best) { best = v; bi = i; bj = j; found = 1; }
            }
            if (status) break;
        }
        if (status) break;
        const size_t c = n + step;
        const double mi = size[bi], mj = size[bj], den = mi + mj;
        for (size_t k = 0; k < d; ++k) /* merge: FastClusterWrapper.cpp:89-100 */
            pts[c * d + k] = (pts[bi * d + k] * mi + pts[bj * d + k] * mj) / den;
        size[c] = den;
        active[bi] = active[bj] = 0;
        active[c] = 1;
        z[4 * step + 0] = (double)(bi < bj ? bi : bj); /* LinkageOutput::append :150-160 */
        z[4 * step + 1] = (double)(bi < bj ? bj : bi);
        z[4 * step + 2] = sqrt(best);                  /* postprocess :128-130 */
        z[4 * step + 3] = den;
    }
    free(pts); free(size); free(active);
    return status;
}

/* ==================================== VBx ========================================= */

int fa_oracle_vbx_run(const double *features, long T, long D, const double *phi,
                      double *gamma, long S, int max_iter, double epsilon,
                      double Fa, double Fb, double init_smoothing,
                      double *pi, double *elbos) {
    /* VBxClustering.swift:167-664 */
    double *row = (double *)malloc(sizeof(double) * (size_t)S);
    if (init_smoothing >= 0.0) { /* :190-219 */
        for (long t = 0; t < T; ++t) {
            double *g = gamma + t * S;
            double mx = -1.7976931348623157e308;
            for (long s = 0; s < S; ++s) { row[s] = g[s] * init_smoothing; if (row[s] > mx) mx = row[s]; }
            double sum = 0.0;
            for (long s = 0; s < S; ++s) { row[s] = exp(row[s] + (-mx)); sum += row[s]; }
            if (sum <= 0.0 || !isfinite(sum)) for (long s = 0; s < S; ++s) g[s] = 1.0 / (double)S;
            else { const double inv = 1.0 / sum; for (long s = 0; s < S; ++s) g[s] = row[s] * inv; }
        }
    }
    for (long t = 0; t < T; ++t) { /* :222-237 */
        double *g = gamma + t * S;
        double sum = 0.0;
        for (long s = 0; s < S; ++s) sum += g[s];
        if (sum <= 0.0 || !isfinite(sum)) for (long s = 0; s < S; ++s) g[s] = 1.0 / (double)S;
        else { const double inv = 1.0 / sum; for (long s = 0; s < S; ++s) g[s] *= inv; }
    }
    for (long s = 0; s < S; ++s) pi[s] = 1.0 / (double)S; /* :239 */
    double *phic = (double *)malloc(sizeof(double) * (size_t)D);
    double *rho = (double *)malloc(sizeof(double) * (size_t)T * (size_t)D);
    double *G = (double *)malloc(sizeof(double) * (size_t)T);
    double *invL = (double *)malloc(sizeof(double) * (size_t)S * (size_t)D);
    double *alpha = (double *)malloc(sizeof(double) * (size_t)S * (size_t)D);
    double *phiT = (double *)malloc(sizeof(double) * (size_t)S);
    double *gsum = (double *)malloc(sizeof(double) * (size_t)S);
    double *logP = (double *)malloc(sizeof(double) * (size_t)T * (size_t)S);
    double *logPi = (double *)malloc(sizeof(double) * (size_t)S);
    for (long d = 0; d < D; ++d) phic[d] = phi[d] > 1e-12 ? phi[d] : 1e-12; /* :241 */
    for (long t = 0; t < T; ++t)
        for (long d = 0; d < D; ++d) rho[t * D + d] = features[t * D + d] * sqrt(phic[d]); /* :242-265 */
    const double log_const = (double)D * log(2.0 * M_PI);
    for (long t = 0; t < T; ++t) { /* :267-282 */
        double ss = 0.0;
        for (long d = 0; d < D; ++d) ss += features[t * D + d] * features[t * D + d];
        G[t] = -0.5 * (ss + log_const);
    }
    const double ratio = Fa / Fb;
    double prev = -1.7976931348623157e308;
    int iters = 0;
    for (int it = 0; it < max_iter; ++it) {
        iters = it + 1;
        for (long s = 0; s < S; ++s) gsum[s] = 0.0; /* dgemv :312-325 */
        for (long t = 0; t < T; ++t) for (long s = 0; s < S; ++s) gsum[s] += gamma[t * S + s];
        for (long s = 0; s < S; ++s) { /* :330-337 */
            const double w = ratio * gsum[s];
            for (long d = 0; d < D; ++d) {
                const double den = 1.0 + w * phic[d];
                invL[s * D + d] = 1.0 / (den > 1e-12 ? den : 1e-12);
            }
        }
        for (long i = 0; i < S * D; ++i) alpha[i] = 0.0; /* dgemm gamma^T rho :342-357 */
        for (long t = 0; t < T; ++t)
            for (long s = 0; s < S; ++s) {
                const double g = gamma[t * S + s];
                if (g == 0.0) continue;
                for (long d = 0; d < D; ++d) alpha[s * D + d] += g * rho[t * D + d];
            }
        for (long i = 0; i < S * D; ++i) alpha[i] = (alpha[i] * invL[i]) * ratio; /* :370-387 */
        for (long s = 0; s < S; ++s) { /* :402-432 */
            double sum = 0.0;
            for (long d = 0; d < D; ++d) sum += (alpha[s * D + d] * alpha[s * D + d] + invL[s * D + d]) * phic[d];
            phiT[s] = sum;
        }
        for (long t = 0; t < T; ++t) /* dgemm rho alpha^T :441-456, then :485-492 */
            for (long s = 0; s < S; ++s) {
                double dot = 0.0;
                for (long d = 0; d < D; ++d) dot += rho[t * D + d] * alpha[s * D + d];
                logP[t * S + s] = ((dot + phiT[s] * -0.5) + G[t]) * Fa;
            }
        for (long s = 0; s < S; ++s) logPi[s] = log(pi[s] >= 1e-8 ? pi[s] : 1e-8); /* :498-514 */
        double ll = 0.0;
        for (long t = 0; t < T; ++t) { /* :516-572 */
            double mx = -1.7976931348623157e308;
            for (long s = 0; s < S; ++s) { row[s] = logP[t * S + s] + logPi[s]; if (row[s] > mx) mx = row[s]; }
            double sum = 0.0;
            for (long s = 0; s < S; ++s) { row[s] = exp(row[s] + (-mx)); sum += row[s]; }
            if (sum <= 0.0 || !isfinite(sum)) {
                for (long s = 0; s < S; ++s) gamma[t * S + s] = 1.0 / (double)S;
                ll += mx;
            } else {
                const double inv = 1.0 / sum;
                for (long s = 0; s < S; ++s) gamma[t * S + s] = row[s] * inv;
                ll += mx + log(sum);
            }
        }
        for (long s = 0; s < S; ++s) pi[s] = 0.0; /* :586-621 */
        for (long t = 0; t < T; ++t) for (long s = 0; s < S; ++s) pi[s] += gamma[t * S + s];
        double psum = 0.0;
        for (long s = 0; s < S; ++s) psum += pi[s];
        if (psum > 0.0 && isfinite(psum)) { const double inv = 1.0 / psum; for (long s = 0; s < S; ++s) pi[s] *= inv; }
        else for (long s = 0; s < S; ++s) pi[s] = 1.0 / (double)S;
        double sli = 0.0, si = 0.0, sa = 0.0; /* :623-647 */
        for (long i = 0; i < S * D; ++i) { sli += log(invL[i]); si += invL[i]; sa += alpha[i] * alpha[i]; }
        const double elbo = ll + Fb * 0.5 * (sli - si - sa + (double)(S * D));
        elbos[it] = elbo;
        if (it > 0 && fabs(elbo - prev) < epsilon) { prev = elbo; break; } /* :653-659 */
        prev = elbo;
    }
    free(row); free(phic); free(rho); free(G); free(invL); free(alpha); free(phiT); free(gsum);
    free(logP); free(logPi);
    return iters;
}

static int cmp_i32(const void *a, const void *b) {
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

int fa_oracle_vbx_refine(const double *rho, long T, long D, const int32_t *initial, const double *phi,
                         int max_iter, double epsilon, double Fa, double Fb,
                         double *gamma, double *pi, int32_t *hard, double *elbos, long *S_out) {
    /* VBxClustering.swift:41-165 */
    if (T <= 0 || D <= 0) { *S_out = 0; return 0; }
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
    memcpy(tmp, initial, sizeof(int32_t) * (size_t)T);
    qsort(tmp, (size_t)T, sizeof(int32_t), cmp_i32);
    long S = 1;
    for (long i = 1; i < T; ++i) if (tmp[i] != tmp[i - 1]) ++S; /* Set(initialClusters).count :78 */
    free(tmp);
    memset(gamma, 0, sizeof(double) * (size_t)T * (size_t)S);
    for (long t = 0; t < T; ++t) { /* :102-107 */
        long sp = initial[t];
        if (sp > S - 1) sp = S - 1;
        if (sp < 0) sp = 0;
        gamma[t * S + sp] = 1.0;
    }
    const int iters = fa_oracle_vbx_run(rho, T, D, phi, gamma, S, max_iter, epsilon, Fa, Fb, 7.0, pi, elbos);
    for (long t = 0; t < T; ++t) { /* :144-146 first max */
        long b = 0;
        for (long s = 1; s < S; ++s) if (gamma[t * S + b] < gamma[t * S + s]) b = s;
        hard[t] = (int32_t)b;
    }
    *S_out = S;
    return iters;
}

/* ================================= post-VBx ======================================= */

long fa_oracle_weighted_centroids(const double *emb, long n, long d, const double *gamma, const double *pi,
                                  long S, double *centroids, int32_t *map) {
    /* OfflineDiarizerManager.swift:630-684 */
    long K = 0;
    for (long s = 0; s < S; ++s) {
        map[s] = -1;
        if (!(pi[s] > 1e-7)) continue;
        double *num = centroids + K * d;
        for (long k = 0; k < d; ++k) num[k] = 0.0;
        double den = 0.0;
        for (long t = 0; t < n; ++t) {
            const double w = gamma[t * S + s];
            if (!(w > 0)) continue;
            den += w;
            for (long k = 0; k < d; ++k) num[k] += w * emb[t * d + k]; /* cblas_daxpy */
        }
        if (den > 0) for (long k = 0; k < d; ++k) num[k] /= den;
        else for (long k = 0; k < d; ++k) num[k] = 0.0;
        map[s] = (int32_t)K++;
    }
    return K;
}

static void normalize_vec(const double *v, long d, double *out) {
    /* OfflineDiarizerManager.swift:824-859: sumSquares <= 0 returns the vector unchanged */
    double ss = 0.0;
    for (long k = 0; k < d; ++k) ss += v[k] * v[k];
    const double scale = ss <= 0 ? 1.0 : 1.0 / sqrt(ss);
    for (long k = 0; k < d; ++k) out[k] = ss <= 0 ? v[k] : v[k] * scale;
}

void fa_oracle_assign_cosine(const double *emb, long n, long d, const double *centroids, long K,
                             int32_t *out) {
    /* OfflineDiarizerManager.swift:789-822 */
    if (K <= 0) { for (long i = 0; i < n; ++i) out[i] = 0; return; }
    double *cn = (double *)malloc(sizeof(double) * (size_t)K * (size_t)d);
    double *e = (double *)malloc(sizeof(double) * (size_t)d);
    for (long k = 0; k < K; ++k) normalize_vec(centroids + k * d, d, cn + k * d);
    for (long i = 0; i < n; ++i) {
        normalize_vec(emb + i * d, d, e);
        double best = -INFINITY;
        int32_t bi = 0;
        for (long k = 0; k < K; ++k) {
            double dot = 0.0;
            for (long j = 0; j < d; ++j) dot += e[j] * cn[k * d + j];
            if (dot > best) { best = dot; bi = (int32_t)k; }
        }
        out[i] = bi;
    }
    free(cn); free(e);
}

/* ================================ resampling ================================== */

/* AudioConverter.swift:420 */
long fa_oracle_resample_linear_frames(long frames, double in_rate, double out_rate) {
    if (in_rate == out_rate) return frames;
    return (long)((double)frames / (in_rate / out_rate));
}

/* AudioConverter.linearResample (AudioConverter.swift:388-442): planar [channels][frames] -> mono -> linear interpolation */
long fa_oracle_resample_linear(const float *planar, int channels, long frames, double in_rate, double out_rate, float *out) {
    float *mono = (float *)malloc(sizeof(float) * (size_t)(frames > 0 ? frames : 1));
    const float weight = 1.0f / (float)channels;                       /* :400 */
    for (long f = 0; f < frames; ++f) {
        float sum = 0.0f;
        for (int c = 0; c < channels; ++c) sum += planar[(size_t)c * frames + f];
        mono[f] = sum * weight;                                         /* :402-408 */
    }
    if (in_rate == out_rate) {                                          /* :414-416 */
        memcpy(out, mono, sizeof(float) * (size_t)frames);
        free(mono);
        return frames;
    }
    const double ratio = in_rate / out_rate;                            /* :419 */
    const long n_out = (long)((double)frames / ratio);                  /* :420 */
    for (long i = 0; i < n_out; ++i) {
        const double src = (double)i * ratio;
        const long idx = (long)src;
        const float frac = (float)(src - (double)idx);
        float v = 0.0f;
        if (idx < frames - 1) v = mono[idx] * (1.0f - frac) + mono[idx + 1] * frac;   /* :428-430 */
        else if (idx < frames) v = mono[idx];                                           /* :431-432 */
        out[i] = v;
    }
    free(mono);
    return n_out;
}

/* UnifiedMelExtractor.normalizePerFeature (ASR/Parakeet/Unified/UnifiedMelExtractor.swift:91-113) on a time-major
 * [frames][n_mels] buffer, in place */
void fa_oracle_normalize_per_feature(float *x, int n_mels, int frames, int valid_frames) {
    if (valid_frames <= 0) { for (long i = 0; i < (long)frames * n_mels; ++i) x[i] = 0.0f; return; }
    const float denom = (float)(valid_frames > 1 ? valid_frames - 1 : 1);
    for (int m = 0; m < n_mels; ++m) {
        float mean = 0.0f;
        for (int t = 0; t < valid_frames; ++t) mean += x[(long)t * n_mels + m];
        mean /= (float)valid_frames;
        float var_sum = 0.0f;
        for (int t = 0; t < valid_frames; ++t) { const float d = x[(long)t * n_mels + m] - mean; var_sum += d * d; }
        const float std = sqrtf(var_sum / denom) + 1e-5f;
        for (int t = 0; t < frames; ++t) x[(long)t * n_mels + m] = t < valid_frames ? (x[(long)t * n_mels + m] - mean) / std : 0.0f;
    }
}

/* ================================ constrained assignment ====================== */

/* HungarianAssignment.solve (Diarizer/HungarianAssignment.swift:8-62): Kuhn-Munkres with potentials, 1-based arrays */
void fa_oracle_hungarian_solve(const long long *cost, int n, int *assign) {
    if (n == 0) return;
    const long long INF = 0x7fffffffffffffffLL / 4;
    long long *u = (long long *)calloc((size_t)n + 1, sizeof(long long)), *v = (long long *)calloc((size_t)n + 1, sizeof(long long));
    long long *minv = (long long *)malloc(sizeof(long long) * ((size_t)n + 1));
    int *p = (int *)calloc((size_t)n + 1, sizeof(int)), *way = (int *)calloc((size_t)n + 1, sizeof(int));
    char *used = (char *)malloc((size_t)n + 1);
    for (int i = 1; i <= n; ++i) {
        p[0] = i;
        int j0 = 0;
        for (int j = 0; j <= n; ++j) { minv[j] = INF; used[j] = 0; }
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            long long delta = INF;
            int j1 = 0;
            for (int j = 1; j <= n; ++j) {
                if (used[j]) continue;
                const long long cur = cost[(size_t)(i0 - 1) * n + (j - 1)] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            for (int j = 0; j <= n; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do { const int j1 = way[j0]; p[j0] = p[j1]; j0 = j1; } while (j0 != 0);
    }
    for (int r = 0; r < n; ++r) assign[r] = -1;
    for (int j = 1; j <= n; ++j) if (p[j] != 0) assign[p[j] - 1] = j - 1;
    free(u); free(v); free(minv); free(p); free(way); free(used);
}

/* HungarianAssignment.maxScoreAssignment (:67-97) on row-major scores[rows][cols] */
void fa_oracle_max_score_assignment(const double *scores, int rows, int cols, int *assign) {
    if (rows <= 0) return;
    if (cols <= 0) { for (int r = 0; r < rows; ++r) assign[r] = -1; return; }
    double mx = -INFINITY, mn = INFINITY;
    for (long i = 0; i < (long)rows * cols; ++i) if (isfinite(scores[i])) { if (scores[i] > mx) mx = scores[i]; if (scores[i] < mn) mn = scores[i]; }
    const double max_score = mx == -INFINITY ? 0.0 : mx, min_score = mn == INFINITY ? 0.0 : mn;
    const double sentinel = min_score - 1.0;
    const int n = rows > cols ? rows : cols;
    long long *cost = (long long *)calloc((size_t)n * n, sizeof(long long));
    int *full = (int *)malloc(sizeof(int) * (size_t)n);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const double sc = isfinite(scores[(long)r * cols + c]) ? scores[(long)r * cols + c] : sentinel;
            cost[(size_t)r * n + c] = (long long)round((max_score - sc) * 1e6);
        }
    fa_oracle_hungarian_solve(cost, n, full);
    for (int r = 0; r < rows; ++r) assign[r] = full[r] < cols ? full[r] : -1;
    free(cost); free(full);
}

/* ConstrainedClusterAssignment.assign (Diarizer/Offline/Clustering/ConstrainedClusterAssignment.swift:20-42) */
void fa_oracle_constrained_assign(const double *scores, long n, int K, const int32_t *chunk, int32_t *out) {
    for (long i = 0; i < n; ++i) out[i] = -2;
    char *done = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
    long *rows = (long *)malloc(sizeof(long) * (size_t)(n > 0 ? n : 1));
    for (long i = 0; i < n; ++i) {
        if (done[i]) continue;
        long cnt = 0;
        for (long j = i; j < n; ++j) if (!done[j] && chunk[j] == chunk[i]) { rows[cnt++] = j; done[j] = 1; }
        double *cs = (double *)malloc(sizeof(double) * (size_t)cnt * (size_t)(K > 0 ? K : 1));
        int *as = (int *)malloc(sizeof(int) * (size_t)cnt);
        for (long r = 0; r < cnt; ++r) for (int c = 0; c < K; ++c) cs[r * K + c] = scores[rows[r] * K + c];
        fa_oracle_max_score_assignment(cs, (int)cnt, K, as);
        for (long r = 0; r < cnt; ++r) out[rows[r]] = as[r] >= 0 ? as[r] : -2;
        free(cs); free(as);
    }
    free(done); free(rows);
}

/* centroidScores (OfflineDiarizerManager.swift:789-798) */
void fa_oracle_centroid_scores(const double *emb, long n, long d, const double *centroids, long K, double *scores) {
    double *cn = (double *)malloc(sizeof(double) * (size_t)(K > 0 ? K : 1) * (size_t)d);
    double *e = (double *)malloc(sizeof(double) * (size_t)d);
    for (long k = 0; k < K; ++k) normalize_vec(centroids + k * d, d, cn + k * d);
    for (long i = 0; i < n; ++i) {
        normalize_vec(emb + i * d, d, e);
        for (long k = 0; k < K; ++k) {
            double dot = 0.0;
            for (long j = 0; j < d; ++j) dot += e[j] * cn[k * d + j];
            scores[i * K + k] = dot;
        }
    }
    free(cn); free(e);
}

/* ================================ TDT control loop ============================ */

int fa_oracle_tdt_initial_time_index(int has_time_jump, int time_jump, int context_frame_adjustment) {
    /* TdtFrameNavigation.swift:20-49 */
    if (!has_time_jump) return context_frame_adjustment;
    if (time_jump == 0 && context_frame_adjustment == 0) return 25; /* ASRConstants.standardOverlapFrames */
    return time_jump + context_frame_adjustment > 0 ? time_jump + context_frame_adjustment : 0;
}

float fa_oracle_tdt_clamp_probability(float v) {
    /* TdtDurationMapping.swift:28-31 */
    if (!isfinite(v)) return 0.0f;
    return fmaxf(0.0f, fminf(1.0f, v));
}

typedef struct {
    const int32_t *tok, *bin; const float *prob; int U, T, u;
    const int *bins; int nbins;
    int token, duration, err; float score;
} tdt_joint;

static __thread long tdt_joint_calls; /* joint evaluations of the calling thread's last fa_oracle_tdt_greedy (what a walk on logits reads: one row each) */
long fa_oracle_tdt_last_joint_calls(void) { return tdt_joint_calls; }

static int tdt_run_joint(tdt_joint *j, int frame) { /* TdtModelInference.runJointPrepared, served from the decision tables */
    if (j->u >= j->U || frame < 0 || frame >= j->T) { j->err = 3; return 0; }
    tdt_joint_calls += 1;
    const long i = (long)j->u * j->T + frame;
    j->token = j->tok[i];
    j->score = fa_oracle_tdt_clamp_probability(j->prob[i]);
    if (j->bin[i] < 0 || j->bin[i] >= j->nbins) { j->err = 5; return 0; } /* mapDurationBin throws (TdtDurationMapping.swift:17-22) */
    j->duration = j->bins[j->bin[i]];
    return 1;
}

/* TdtDecoderV3.decodeWithTimings (TdtDecoderV3.swift:103-607) for ONE chunk with language == nil; the joint decisions come
 * from tables [U][T] indexed by (decoder steps taken, encoder frame).  Returns the status (0 ok, 5 duration bin out of
 * range, 3 table/output exhausted); *final_time == INT32_MIN when the reference returns before updating timeJump. */
int fa_oracle_tdt_greedy(const int32_t *tok, const int32_t *bin, const float *prob, int U, int T, int enc_len, int audio_frames,
                         int t0, int is_last, int global_offset, int emit_after, int blank_id, int max_symbols, int max_tokens,
                         int blank_limit, const int *bins, int nbins, int max_out, int32_t *out_tok, int32_t *out_time,
                         int32_t *out_dur, float *out_conf, int *out_count, int *final_time, int *final_u) {
    tdt_joint j = {tok, bin, prob, U, T, 0, bins, nbins, blank_id, 0, 0, 0.0f};
    int count = 0, status = 0;
    tdt_joint_calls = 0;
    *out_count = 0; *final_u = 0; *final_time = INT32_MIN;
    if (enc_len <= 1) return 0;                                         /* :110-112 */
    int time_indices = t0;
    const int effective = enc_len < audio_frames ? enc_len : audio_frames;
    int safe = time_indices < effective - 1 ? time_indices : effective - 1;
    const int last_timestep = effective - 1;
    int active = time_indices < effective;
    int time_indices_current_labels = time_indices;
    if (time_indices >= effective) return 0;                            /* :150-152 */
    int last_emission_timestamp = -1, emissions_at_this_timestamp = 0, tokens_processed = 0;
    int label = blank_id, duration = 0; float score = 0.0f;
#define TDT_EMIT(ts)                                                                                         \
    do {                                                                                                     \
        if (emit_after < 0 || (ts) >= emit_after) {                                                          \
            if (count < max_out) { out_tok[count] = label; out_time[count] = (ts); out_dur[count] = duration; out_conf[count] = score; } \
            else status = 3;                                                                                 \
            ++count;                                                                                         \
        }                                                                                                    \
    } while (0)
    while (active) {                                                    /* :230 */
        if (!tdt_run_joint(&j, safe)) { status = j.err; goto done; }
        label = j.token; score = j.score; duration = j.duration;
        int blank_mask = label == blank_id;
        const int current_time_index = time_indices;
        if (!blank_mask && duration == 0 && current_time_index == last_emission_timestamp && emissions_at_this_timestamp >= 1) duration = 1;
        if (blank_mask && duration == 0) duration = 1;
        time_indices_current_labels = time_indices;
        time_indices += duration;
        safe = time_indices < last_timestep ? time_indices : last_timestep;
        active = time_indices < effective;
        int advance = active && blank_mask;
        while (advance) {                                               /* :348-405 */
            time_indices_current_labels = time_indices;
            if (!tdt_run_joint(&j, safe)) { status = j.err; goto done; }
            label = j.token; score = j.score; duration = j.duration;
            blank_mask = label == blank_id;
            if (blank_mask && duration == 0) duration = 1;
            time_indices += duration;
            safe = time_indices < last_timestep ? time_indices : last_timestep;
            active = time_indices < effective;
            advance = active && blank_mask;
        }
        if (active && label != blank_id) {                              /* :409 */
            tokens_processed += 1;
            if (tokens_processed > max_tokens) break;
            TDT_EMIT(time_indices_current_labels + global_offset);
            j.u += 1;                                                   /* runDecoder(token) (:433-444) */
            if (time_indices_current_labels == last_emission_timestamp) emissions_at_this_timestamp += 1;
            else { last_emission_timestamp = time_indices_current_labels; emissions_at_this_timestamp = 1; }
            if (emissions_at_this_timestamp >= max_symbols) {
                time_indices = time_indices + 1 < last_timestep ? time_indices + 1 : last_timestep;
                safe = time_indices < last_timestep ? time_indices : last_timestep;
                emissions_at_this_timestamp = 0;
                last_emission_timestamp = -1;
            }
        }
        active = time_indices < effective;
    }
    if (is_last) {                                                      /* :472-571 */
        int additional = 0, consecutive_blanks = 0, fp = time_indices;
        while (additional < max_symbols && consecutive_blanks < blank_limit) {
            int var[3];
            var[0] = fp < enc_len - 1 ? fp : enc_len - 1;
            var[1] = effective - 1 < enc_len - 1 ? effective - 1 : enc_len - 1;
            var[2] = (effective - 2 > 0 ? effective - 2 : 0) < enc_len - 1 ? (effective - 2 > 0 ? effective - 2 : 0) : enc_len - 1;
            if (!tdt_run_joint(&j, var[additional % 3])) { status = j.err; goto done; }
            label = j.token; score = j.score; duration = j.duration;
            if (label == blank_id) consecutive_blanks += 1;
            else {
                consecutive_blanks = 0;
                const int final_ts = (fp < effective - 1 ? fp : effective - 1) + global_offset;
                TDT_EMIT(final_ts);
                j.u += 1;
            }
            const int adv = duration > 1 ? duration : 1;
            fp = fp + adv < effective ? fp + adv : effective;
            additional += 1;
        }
    }
    *final_time = time_indices;
done:
    *out_count = count; *final_u = j.u;
    return status;
#undef TDT_EMIT
}

/* CtcKeywordSpotter.logSoftmax + blank bias (CtcKeywordSpotter+Inference.swift:397-431), one row */
void fa_oracle_log_softmax_row(const float *logits, int V, float temperature, float blank_bias, int blank_id, float *out) {
    if (V <= 0) return;
    float mx = -INFINITY;
    for (int i = 0; i < V; ++i) { const float v = temperature != 1.0f ? logits[i] / temperature : logits[i]; out[i] = v; if (v > mx) mx = v; }
    float sum = 0.0f;
    for (int i = 0; i < V; ++i) sum += expf(out[i] - mx);
    const float lse = logf(sum);
    for (int i = 0; i < V; ++i) out[i] = (out[i] - mx) - lse;
    if (blank_bias != 0.0f && blank_id >= 0 && blank_id < V) out[blank_id] -= blank_bias;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * K-Means fallback for speaker-count constraints (KMeansClustering.swift:39-224) + SpeakerCountConstraints.
 *
 * The random draws come from two places: the reference's own LCG (SeededRNG.next, KMeansClustering.swift:212-223) and the
 * SWIFT STANDARD LIBRARY (third party, not under /root/reference, toolchain version unpinned by Package.swift):
 *   - RandomNumberGenerator.next(upperBound:)  — Lemire's "nearly divisionless" method on the full-width product
 *     (stdlib/public/core/Random.swift, Swift >= 5.0): m = next() * bound (128 bit); if low64(m) < bound:
 *     t = (0 - bound) % bound; while low64(m) < t: redraw; return high64(m).
 *   - MutableCollection.shuffle(using:) (CollectionAlgorithms.swift): for amount = count, count-1, ..., 2:
 *     swapAt(cur, cur + Int.random(in: 0..<amount)); cur += 1.
 *   - Collection.randomElement(using:): self[Int.random(in: 0..<count)].
 * The reference tests pin only structural outcomes for fixed seeds (KMeansClusteringTests.swift:10-131), which the
 * restatement reproduces (tests/test_oracle_kmeans.py); draw-level parity with a real Swift toolchain is UNPINNED.
 * vDSP_svesqD's internal summation order is not documented; sums here are sequential in the dimension.
 * ------------------------------------------------------------------------------------------------------------------- */
uint64_t fa_oracle_seeded_rng_next(uint64_t *state) {                    /* KMeansClustering.swift:219-222 */
    *state = *state * 6364136223846793005ULL + 1442695040888963407ULL;
    return *state;
}

uint64_t fa_oracle_rng_upper_bound(uint64_t *state, uint64_t bound) {    /* Swift stdlib next(upperBound:) */
    uint64_t r = fa_oracle_seeded_rng_next(state);
    unsigned __int128 m = (unsigned __int128)r * bound;
    if ((uint64_t)m < bound) {
        const uint64_t t = (0 - bound) % bound;
        while ((uint64_t)m < t) {
            r = fa_oracle_seeded_rng_next(state);
            m = (unsigned __int128)r * bound;
        }
    }
    return (uint64_t)(m >> 64);
}

void fa_oracle_shuffle_indices(uint64_t *state, long n, int64_t *idx) {  /* Swift stdlib shuffle(using:) */
    for (long i = 0; i < n; ++i) idx[i] = i;
    long amount = n, cur = 0;
    while (amount > 1) {
        const long r = (long)fa_oracle_rng_upper_bound(state, (uint64_t)amount);
        amount -= 1;
        const int64_t tmp = idx[cur]; idx[cur] = idx[cur + r]; idx[cur + r] = tmp;
        cur += 1;
    }
}

static double km_dist2(const double *a, const double *b, long d) {       /* euclideanDistanceSquared :170-177 */
    double s = 0.0;
    for (long k = 0; k < d; ++k) { const double df = a[k] - b[k]; s += df * df; }
    return s;
}

void fa_oracle_kmeans_normalize(const double *x, long n, long d, double *out) {   /* normalizeEmbeddings :131-142 */
    for (long i = 0; i < n; ++i) {
        double ss = 0.0;
        for (long k = 0; k < d; ++k) ss += x[i * d + k] * x[i * d + k];
        const double norm = sqrt(ss);
        if (!(norm > 1e-10)) { memcpy(out + i * d, x + i * d, (size_t)d * sizeof(double)); continue; }
        const double inv = 1.0 / norm;
        for (long k = 0; k < d; ++k) out[i * d + k] = x[i * d + k] * inv;
    }
}

/* clusterWithCentroids (:39-91).  labels[n]; centroids[min(k,n)*d] (may be NULL); *out_k = number of centroid rows
 * written (0 for the degenerate returns that carry no centroids); *out_iters = assignment passes executed. */
int fa_oracle_kmeans(const double *emb, long n, long d, long num_clusters, long max_iter, uint64_t seed,
                     int32_t *labels, double *centroids, long *out_k, long *out_iters) {
    if (out_k) *out_k = 0;
    if (out_iters) *out_iters = 0;
    if (n <= 0) return 0;                                                /* :46-48 */
    if (d <= 0) { for (long i = 0; i < n; ++i) labels[i] = 0; return 0; }   /* :49-51 */
    const long k = num_clusters < n ? num_clusters : n;
    if (k <= 0) { for (long i = 0; i < n; ++i) labels[i] = 0; return 0; }   /* :54-56 */
    if (n <= k) {                                                        /* :57-59: identity labels, raw embeddings */
        for (long i = 0; i < n; ++i) labels[i] = (int32_t)i;
        if (centroids) memcpy(centroids, emb, (size_t)(n * d) * sizeof(double));
        if (out_k) *out_k = n;
        return 0;
    }
    uint64_t rng = seed;
    double *xn = malloc((size_t)(n * d) * sizeof(double));
    double *cen = malloc((size_t)(k * d) * sizeof(double));
    double *sums = malloc((size_t)(k * d) * sizeof(double));
    int64_t *idx = malloc((size_t)n * sizeof(int64_t));
    long *cnt = malloc((size_t)k * sizeof(long));
    int32_t *cur = calloc((size_t)n, sizeof(int32_t)), *nxt = malloc((size_t)n * sizeof(int32_t));
    if (!xn || !cen || !sums || !idx || !cnt || !cur || !nxt) { free(xn); free(cen); free(sums); free(idx); free(cnt); free(cur); free(nxt); return 4; }
    fa_oracle_kmeans_normalize(emb, n, d, xn);
    fa_oracle_shuffle_indices(&rng, n, idx);                             /* initializeCentroids :144-152 */
    for (long c = 0; c < k; ++c) memcpy(cen + c * d, xn + idx[c] * d, (size_t)d * sizeof(double));
    long it = 0;
    for (; it < max_iter; ++it) {
        int same = 1;
        for (long i = 0; i < n; ++i) {                                   /* assignToCentroids :154-168 */
            int best = 0; double bd = DBL_MAX;
            for (long c = 0; c < k; ++c) { const double ds = km_dist2(xn + i * d, cen + c * d, d); if (ds < bd) { bd = ds; best = (int)c; } }
            nxt[i] = best; if (best != cur[i]) same = 0;
        }
        if (same) { it += 1; break; }                                    /* :70-73 */
        memcpy(cur, nxt, (size_t)n * sizeof(int32_t));
        memset(sums, 0, (size_t)(k * d) * sizeof(double));               /* updateCentroids :179-207 */
        memset(cnt, 0, (size_t)k * sizeof(long));
        for (long i = 0; i < n; ++i) { const long c = cur[i]; cnt[c] += 1; for (long q = 0; q < d; ++q) sums[c * d + q] += xn[i * d + q]; }
        for (long c = 0; c < k; ++c) {
            if (cnt[c] == 0) {                                           /* empty cluster: random data point (:196-199) */
                const long r = (long)fa_oracle_rng_upper_bound(&rng, (uint64_t)n);
                memcpy(cen + c * d, xn + r * d, (size_t)d * sizeof(double));
            } else {
                const double inv = 1.0 / (double)cnt[c];
                for (long q = 0; q < d; ++q) cen[c * d + q] = sums[c * d + q] * inv;
            }
        }
    }
    memcpy(labels, cur, (size_t)n * sizeof(int32_t));
    if (centroids) memcpy(centroids, cen, (size_t)(k * d) * sizeof(double));
    if (out_k) *out_k = k;
    if (out_iters) *out_iters = it;
    free(xn); free(cen); free(sums); free(idx); free(cnt); free(cur); free(nxt);
    return 0;
}

/* clusterWithCentroidsNInit (:99-129): seeds base, base+1, ...; lowest inertia, first on ties. */
int fa_oracle_kmeans_ninit(const double *emb, long n, long d, long num_clusters, long max_iter, long n_init, uint64_t base_seed,
                           int32_t *labels, double *centroids, long *out_k, long *best_run, double *inertias) {
    if (best_run) *best_run = 0;
    if (!(n > num_clusters && n_init > 1))                               /* :106-110 */
        return fa_oracle_kmeans(emb, n, d, num_clusters, max_iter, base_seed, labels, centroids, out_k, NULL);
    const long kmax = num_clusters < n ? (num_clusters > 0 ? num_clusters : 0) : n;
    double *xn = malloc((size_t)(n * (d > 0 ? d : 1)) * sizeof(double));
    double *cen = malloc((size_t)((kmax > 0 ? kmax : 1) * (d > 0 ? d : 1)) * sizeof(double));
    int32_t *lab = malloc((size_t)n * sizeof(int32_t));
    if (!xn || !cen || !lab) { free(xn); free(cen); free(lab); return 4; }
    if (d > 0) fa_oracle_kmeans_normalize(emb, n, d, xn);
    double best = DBL_MAX; int have = 0;
    for (long r = 0; r < n_init; ++r) {
        long kk = 0;
        const int st = fa_oracle_kmeans(emb, n, d, num_clusters, max_iter, base_seed + (uint64_t)r, lab, cen, &kk, NULL);
        if (st) { free(xn); free(cen); free(lab); return st; }
        double inertia = 0.0;                                            /* :118-121 */
        for (long i = 0; i < n; ++i) if (lab[i] >= 0 && lab[i] < kk) inertia += km_dist2(xn + i * d, cen + lab[i] * d, d);
        if (inertias) inertias[r] = inertia;
        if (inertia < best) {
            best = inertia; have = 1;
            memcpy(labels, lab, (size_t)n * sizeof(int32_t));
            if (centroids && kk > 0) memcpy(centroids, cen, (size_t)(kk * d) * sizeof(double));
            if (out_k) *out_k = kk;
            if (best_run) *best_run = r;
        }
    }
    free(xn); free(cen); free(lab);
    if (!have) return fa_oracle_kmeans(emb, n, d, num_clusters, max_iter, base_seed, labels, centroids, out_k, NULL);   /* :126-128 */
    return 0;
}

/* SpeakerCountConstraints.resolve (SpeakerCountConstraints.swift:25-62).  has_* = 0 encodes nil.
 * out = {numSpeakers or -1 for nil, minSpeakers, maxSpeakers}. */
void fa_oracle_speaker_constraints(long num_embeddings, int has_num, long num, int has_min, long mn, int has_max, long mx, long out[3]) {
    long rmin = has_num ? num : (has_min ? mn : 1);
    rmin = rmin < num_embeddings ? rmin : num_embeddings; rmin = rmin > 1 ? rmin : 1;
    long rmax = has_num ? num : (has_max ? mx : num_embeddings);
    rmax = rmax < num_embeddings ? rmax : num_embeddings; rmax = rmax > 1 ? rmax : 1;
    if (rmin > rmax) rmin = rmax;
    out[0] = rmin == rmax ? rmin : (has_num ? num : -1);
    out[1] = rmin; out[2] = rmax;
}
