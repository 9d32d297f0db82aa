// mel_generic.h — STFT -> |X|^p -> mel -> log for ANY power-of-two n_fft (64..2048) and the variants the tuned
// n_fft = 512 kernels of mel.hip do not cover: magnitude spectra (power 1), reflect padding, caller-supplied / HTK
// filterbanks, replicated tail frames.  Needed by
//   * AudioMelSpectrogram with a metadata-driven nFFT — LS-EEND derives nFFT = nextPow2(winLength)
//     (Sources/FluidAudio/Diarizer/LS-EEND/LSEENDTypes.swift:55-57, LSEENDPreprocessor.swift:70-82);
//   * the torchaudio-flavoured front end of LuxTtsMelExtractor.extract (Sources/FluidAudio/TTS/LuxTts/
//     LuxTtsMelExtractor.swift:52-132: periodic Hann of n_fft, reflect pad n_fft/2, magnitude, HTK no-norm bank,
//     log(max(v, 1e-7)), lhotse frame count with the last frame replicated) — the only mel path of the reference with
//     a golden vector in its tests, so the DEVICE can be run on it.
// One wavefront per frame, four frames per workgroup: the windowed frame goes to LDS as M = n_fft/2 complex points
// z[m] = x[2m] + i x[2m+1], log2(M) radix-2 Stockham passes ping-pong between two LDS buffers (natural order out, no bit
// reversal), the even/odd recombination yields the n_fft/2 + 1 bins, a sparse filterbank row per lane finishes the frame.
// Throughput is not the point here (the batched NeMo configuration takes the tuned kernels); every sample is still read
// from HBM only ~win/hop times through L2 and every output written once.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fa {
namespace melgen {

constexpr int kWaves = 4;
constexpr int kThreads = 64 * kWaves;

struct GenArgs {
    const float *pcm;
    const int64_t *offsets;      // B + 1
    const int32_t *frames;       // B: frames T(b) the caller gets (expectedFrameCount override applied)
    const int32_t *stft_frames;  // B: frames the signal itself yields (tail_replicate: frames beyond repeat the last of these)
    const float *last;           // B or nullptr
    float *out;
    int32_t *lengths;            // B or nullptr
    const float *window;         // [win]
    const float2 *tw;            // [n_fft/2 + 1]  exp(-2 pi i k / n_fft)
    const int32_t *mel_lo;       // [n_mels] first bin of the row's support
    const int32_t *mel_cnt;      // [n_mels] bins in the support
    const int32_t *mel_start;    // [n_mels] offset of the row's weights in mel_w
    const float *mel_w;
    int64_t utt_stride;
    int32_t batch, frame_stride, n_mels, n_fft, log2_m, win, off, hop, pad;
    float preemph, log_floor;
    int32_t floor_clamped, reflect, magnitude, tail_replicate, frame_major;
};

// pre-emphasised sample i of the utterance (AudioMelSpectrogram.swift:211,:219-225); outside [0, len): 0, or the reflected
// sample (LuxTtsMelExtractor.swift:60-66: left audio[min(-i, n-1)], right audio[max(2n-2-i, 0)])
__device__ __forceinline__ float gen_sample(const float *x, const int64_t len, int64_t i, const float lastv, const float p, const bool reflect) {
    if (i < 0 || i >= len) {
        if (!reflect || len <= 0) return 0.0f;
        if (i < 0) { i = -i; if (i > len - 1) i = len - 1; }
        else { i = 2 * len - 2 - i; if (i < 0) i = 0; }
    }
    const float v = x[i];
    if (p == 0.0f) return v;                               // :363-371: plain copy
    return v - p * (i > 0 ? x[i - 1] : lastv);
}

__global__ __launch_bounds__(kThreads) void mel_generic_kernel(const GenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int N = a.n_fft, M = N >> 1;
    float *bufA = smem + static_cast<size_t>(w) * (2 * N + 8);
    float *bufB = bufA + N + 4;
    const int64_t items = static_cast<int64_t>(a.batch) * a.frame_stride;
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kWaves; base < items; base += static_cast<int64_t>(gridDim.x) * kWaves) {
        const int64_t item = base + w;
        const bool live = item < items;
        const int b = live ? static_cast<int>(item / a.frame_stride) : 0;
        const int t = live ? static_cast<int>(item - static_cast<int64_t>(b) * a.frame_stride) : 0;
        const int T = live ? a.frames[b] : 0;
        const bool compute = live && t < T;
        const int64_t o0 = a.offsets[b];
        const int64_t len = a.offsets[b + 1] - o0;
        const float *x = a.pcm + o0;
        if (live && t == 0 && lane == 0 && a.lengths) a.lengths[b] = T;
        if (compute) {
            int tt = t;
            if (a.tail_replicate) { const int s = a.stft_frames[b]; if (tt > s - 1) tt = s > 0 ? s - 1 : 0; }
            const float lastv = a.last ? a.last[b] : 0.0f;
            const int64_t f0 = static_cast<int64_t>(tt) * a.hop - a.pad;   // signal index of frame position 0 (:234: frame[j] = padded[t hop + j])
            for (int n = lane; n < N; n += 64) {
                float v = 0.0f;
                if (n >= a.off && n < a.off + a.win) v = gen_sample(x, len, f0 + n, lastv, a.preemph, a.reflect != 0) * a.window[n - a.off];
                bufA[n] = v;
            }
        }
        __syncthreads();
        float2 *in = reinterpret_cast<float2 *>(bufA), *outb = reinterpret_cast<float2 *>(bufB);
        const int half = M >> 1;
        for (int s = 0; s < a.log2_m; ++s) {        // radix-2 Stockham pass: p = 2^s
            const int p = 1 << s;
            if (compute) {
                for (int i = lane; i < half; i += 64) {
                    const int k = i & (p - 1);
                    const int j = ((i - k) << 1) + k;
                    const float2 tw = a.tw[static_cast<size_t>(k) * (M >> s)];   // exp(-i pi k / p) = tw[k M / p]
                    const float2 u0 = in[i], u1r = in[i + half];
                    const float2 u1 = make_float2(u1r.x * tw.x - u1r.y * tw.y, u1r.x * tw.y + u1r.y * tw.x);
                    outb[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
                    outb[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
                }
            }
            __syncthreads();
            float2 *tmp = in; in = outb; outb = tmp;
        }
        // `in` holds Z[0..M) in natural order; bins k = 0..M into the other buffer (as floats)
        float *P = reinterpret_cast<float *>(outb);
        if (compute) {
            for (int k = lane; k <= M; k += 64) {
                const float2 zk = in[k & (M - 1)], zc = in[(M - k) & (M - 1)];   // A = Z[k], B = conj(Z[M - k])
                const float er = 0.5f * (zk.x + zc.x), ei = 0.5f * (zk.y - zc.y);   // E = (A + B) / 2
                const float dr = zk.x - zc.x, di = zk.y + zc.y;                    // A - B
                const float orr = 0.5f * di, oi = -0.5f * dr;                       // O = (A - B) / (2 i)
                const float2 tw = a.tw[k];
                const float xr = er + (tw.x * orr - tw.y * oi), xi = ei + (tw.x * oi + tw.y * orr);
                const float pw = xr * xr + xi * xi;
                P[k] = a.magnitude ? sqrtf(pw) : pw;
            }
        }
        __syncthreads();
        if (live) {
            float *ob = a.out + static_cast<int64_t>(b) * a.utt_stride;
            for (int m = lane; m < a.n_mels; m += 64) {
                float val = 0.0f;                                                // padValue for t >= T (:39)
                if (compute) {
                    const float *pp = P + a.mel_lo[m];
                    const float *ww = a.mel_w + a.mel_start[m];
                    float acc = 0.0f;
                    for (int j = 0; j < a.mel_cnt[m]; ++j) acc += ww[j] * pp[j];
                    val = a.floor_clamped ? logf(fmaxf(acc, a.log_floor)) : logf(acc + a.log_floor);   // :542-549
                }
                if (a.frame_major) ob[static_cast<int64_t>(t) * a.n_mels + m] = val;
                else ob[static_cast<int64_t>(m) * a.frame_stride + t] = val;
            }
        }
        __syncthreads();
    }
}

}  // namespace melgen
}  // namespace fa
