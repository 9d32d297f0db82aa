// v_mfma_f64_16x16x4_f64 issue rate on gfx950 (diagnostics): 16 independent accumulator tiles per wavefront, 1 / 2 / 4 wavefronts per SIMD,
// and the same stream with 8 LDS reads per 16 MFMAs (the Gram kernel's inner loop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kIters = 512;
// LDS: 0 registers only | 1 eight ds_read_b64 in front of the 16 MFMAs (the Gram loop) | 2 the same with the accumulators in AccVGPRs
// | 3 four ds_read_b128 (same bytes) | 4 reads of the NEXT step issued in front of the MFMAs of this one (register double buffer)
template <int LDS>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    __shared__ double sm[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 144; i += 256) (&sm[0][0][0])[i] = seed + i;
    __syncthreads();
    v4f64 acc[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = v4f64{0, 0, 0, 0};
    double a[4] = {seed, seed + 1, seed + 2, seed + 3}, b[4] = {seed, seed - 1, seed - 2, seed - 3};
    double an[4] = {0, 0, 0, 0}, bn[4] = {0, 0, 0, 0};
    if (LDS == 4) { for (int t = 0; t < 4; ++t) { a[t] = sm[0][lane >> 4][16 * t + (lane & 15)]; b[t] = sm[1][lane >> 4][16 * t + (lane & 15)]; } }
    for (int it = 0; it < kIters; ++it) {
        const int kr = (it & 3) * 4 + (lane >> 4);
        if (LDS == 1 || LDS == 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] = sm[0][kr][16 * t + (lane & 15)]; b[t] = sm[1][kr][16 * t + (lane & 15)]; }
        }
        if (LDS == 3) {
            typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const d2 x = *reinterpret_cast<const d2 *>(&sm[0][kr][32 * t + 2 * (lane & 15)]), y = *reinterpret_cast<const d2 *>(&sm[1][kr][32 * t + 2 * (lane & 15)]);
                a[2 * t] = x.x; a[2 * t + 1] = x.y; b[2 * t] = y.x; b[2 * t + 1] = y.y;
            }
        }
        if (LDS == 4) {
            const int kn = ((it + 1) & 3) * 4 + (lane >> 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { an[t] = sm[0][kn][16 * t + (lane & 15)]; bn[t] = sm[1][kn][16 * t + (lane & 15)]; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (LDS == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[r][c]) : "v"(a[r]), "v"(b[c]));
                else acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r], b[c], acc[r][c], 0, 0, 0);
            }
        if (LDS == 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t] = an[t]; b[t] = bn[t]; }
        }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
    if (s == 12345.678) out[0] = s;
}
// two operand register sets, loop unrolled by two, ds_read_b128; SPREAD: the four reads of the next step sit between the MFMAs of this one
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool SPREAD>
__global__ __launch_bounds__(256) void k2(double *out, double seed) {
    __shared__ double sm[2][16][144];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 16 * 144; i += 256) (&sm[0][0][0])[i] = seed + i;
    __syncthreads();
    v4f64 acc[4][4];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) acc[r][c] = v4f64{0, 0, 0, 0};
    d2 A[2][2], B[2][2];
    auto rd = [&](const int set, const int t, const int kr) {
        A[set][t] = *reinterpret_cast<const d2 *>(&sm[0][kr][32 * t + 2 * (lane & 15)]);
        B[set][t] = *reinterpret_cast<const d2 *>(&sm[1][kr][32 * t + 2 * (lane & 15)]);
    };
    rd(0, 0, lane >> 4); rd(0, 1, lane >> 4);
    for (int it = 0; it < kIters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kn = ((it + h + 1) & 3) * 4 + (lane >> 4);
            if (!SPREAD) { rd(h ^ 1, 0, kn); rd(h ^ 1, 1, kn); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double av = (r & 1) ? A[h][r >> 1].y : A[h][r >> 1].x, bv = (c & 1) ? B[h][c >> 1].y : B[h][c >> 1].x;
                    acc[r][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[r][c], 0, 0, 0);
                }
                if (SPREAD && r == 0) { A[h ^ 1][0] = *reinterpret_cast<const d2 *>(&sm[0][kn][2 * (lane & 15)]); __builtin_amdgcn_sched_barrier(0); }
                if (SPREAD && r == 1) { B[h ^ 1][0] = *reinterpret_cast<const d2 *>(&sm[1][kn][2 * (lane & 15)]); __builtin_amdgcn_sched_barrier(0); }
                if (SPREAD && r == 2) { A[h ^ 1][1] = *reinterpret_cast<const d2 *>(&sm[0][kn][32 + 2 * (lane & 15)]); __builtin_amdgcn_sched_barrier(0); }
                if (SPREAD && r == 3) { B[h ^ 1][1] = *reinterpret_cast<const d2 *>(&sm[1][kn][32 + 2 * (lane & 15)]); __builtin_amdgcn_sched_barrier(0); }
            }
        }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) s += acc[r][c][0] + acc[r][c][1] + acc[r][c][2] + acc[r][c][3];
    if (s == 12345.678) out[0] = s;
}
template <class K>
float run(K kern, int blocks) {
    double *d;
    (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}
int main() {
    const char *name[7] = {"b128, two register sets, reads in front", "b128, two register sets, reads between the MFMAs", "registers only", "8 ds_read_b64 in front", "same, accumulators in AccVGPRs", "4 ds_read_b128 in front", "reads of the next step in front (double buffer)"};
    for (int wg = 1; wg <= 2; ++wg) {   // workgroups of 4 wavefronts per CU = wavefronts per SIMD
        const float m[7] = {run(k2<false>, 256 * wg), run(k2<true>, 256 * wg), run(k<0>, 256 * wg), run(k<1>, 256 * wg), run(k<2>, 256 * wg), run(k<3>, 256 * wg), run(k<4>, 256 * wg)};
        const double mf = 256.0 * wg * 4 * kIters * 16;   // MFMA instructions
        for (int v = 0; v < 7; ++v)
            printf("%d wavefront(s) per SIMD, %-48s %.3f ms = %5.1f TFLOP/s, %5.1f clk per MFMA and SIMD at 2.4 GHz\n", wg, name[v], m[v], mf * 2048 / m[v] * 1e-9, m[v] * 1e-3 * 2.4e9 / (mf / 1024.0));
    }
    return 0;
}
