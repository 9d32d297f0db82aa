// Ground-truth peaks with hipEvents (diagnostics): packed / scalar fp32 FMA rate and LDS read / write bandwidth, all CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;
template <int OP>
__global__ __launch_bounds__(1024) void valu(float *out, float seed) {
    f2 a[8], b = {seed, seed * 0.5f}, c = {0.25f, 0.125f};
    float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{seed + i, seed - i}; s[i] = seed + i; }
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
                else if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (acc == 12345.678f) out[0] = acc;
}
template <int OP>
__global__ __launch_bounds__(1024) void lds(float *out, int zero) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += blockDim.x) sm[i] = i;
    __syncthreads();
    f2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f2{1.0f * i, 2.0f * i};
    f2 *p = reinterpret_cast<f2 *>(sm) + (tid & 1023) + zero;
    for (int it = 0; it < kIters / 4; ++it) {
        if (OP == 0) {
#pragma unroll
            for (int n = 0; n < 8; ++n) v[n] = p[n * 1024];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(v[n]));
        } else {
#pragma unroll
            for (int n = 0; n < 8; ++n) p[n * 1024] = v[n];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += v[i].x + v[i].y;
    if (acc == 12345.678f) out[0] = acc;
}
template <class K, class... A>
float run(K k, int blocks, int threads, size_t ldsb, A... args) {
    float *d;
    (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), ldsb, 0, d, args...);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), ldsb, 0, d, args...);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}
int main() {
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int threads = 64 * waves;
        const double n = 256.0 * threads * kIters * 16.0;   // lane-instructions
        float m0 = run(valu<0>, 256, threads, 0, 1.0f), m1 = run(valu<1>, 256, threads, 0, 1.0f), m2 = run(valu<2>, 256, threads, 0, 1.0f);
        printf("%2d waves/CU: v_fma_f32 %.1f TFLOP/s | v_pk_fma_f32 %.1f TFLOP/s | v_pk_add_f32 %.1f Tadd/s  (ms %.3f %.3f %.3f)\n", waves, n * 2 / m0 * 1e-9, n * 4 / m1 * 1e-9, n * 2 / m2 * 1e-9, m0, m1, m2);
    }
    for (int waves = 4; waves <= 16; waves *= 2) {
        const int threads = 64 * waves;
        const double bytes = 256.0 * threads * (kIters / 4) * 8.0 * 8.0;
        float r = run(lds<0>, 256, threads, 65536, 0), w = run(lds<1>, 256, threads, 65536, 0);
        printf("%2d waves/CU: ds_read_b64 %.1f TB/s (%.1f B/clk/CU at 2.3 GHz) | ds_write_b64 %.1f TB/s (%.1f B/clk/CU)\n", waves, bytes / r * 1e-9, bytes / r * 1e-9 * 1e12 / 256 / 2.3e9,
               bytes / w * 1e-9, bytes / w * 1e-9 * 1e12 / 256 / 2.3e9);
    }
    return 0;
}
