// Round 5 microbenchmark: plain v_fmac_f32 with a SCALAR tap operand whose taps stream through the scalar cache from a table of S bytes (the decimation tiles: 508 B;
// a rows kernel with wave-uniform phases would need 20 - 40 KB).  Every wavefront walks its own sequence of 1 KB slices (256 taps = one unit of four phases), 32
// taps per block, the NEXT block's taps requested before the present block's multiply-adds.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 smem_taps.hip -o smem_taps && ./smem_taps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef const float __attribute__((address_space(4))) *c_f32;

__device__ __forceinline__ void fmac_s(float &acc, const float tap, const float x) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "s"(tap), "v"(x)); }

template <int AHEAD>
__global__ __launch_bounds__(256) void k(const float *table, int table_floats, float *out, int units) {
    const int wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    float acc0 = threadIdx.x * 1e-9f, acc1 = 1.0f, x = 1.0001f + threadIdx.x * 1e-7f;
    asm volatile("" : "+v"(x));
    for (int u = 0; u < units; ++u) {
        const int slice = (wave * 7 + u * 13) % (table_floats / 256);       // this wavefront's unit: 256 taps
        c_f32 base = (c_f32) table + slice * 256;
        float t[2][32];
        if (AHEAD) {
#pragma unroll
            for (int i = 0; i < 32; ++i) t[0][i] = base[i];
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (AHEAD) {
                if (b + 1 < 8) {
#pragma unroll
                    for (int i = 0; i < 32; ++i) t[(b + 1) & 1][i] = base[32 * (b + 1) + i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 32; ++i) t[b & 1][i] = base[32 * b + i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 32; i += 2) { fmac_s(acc0, t[b & 1][i], x); fmac_s(acc1, t[b & 1][i + 1], x); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1;
}

int main() {
    float *d, *tab;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    hipMalloc(&tab, 1 << 20);
    hipMemset(tab, 0, 1 << 20);
    for (int ahead : {0, 1})
        for (int kb : {1, 8, 16, 20, 40, 160})
            for (int wgs_per_cu : {2, 4}) {                                  // workgroups of 4 wavefronts: 2 / 4 wavefronts per SIMD
                const int units = 400, grid = 256 * wgs_per_cu;
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                auto launch = [&](int n) { if (ahead) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, tab, kb * 256, d, n); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, tab, kb * 256, d, n); };
                launch(4);
                hipEventRecord(e0);
                launch(units);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                const double fmacs_per_simd = double(units) * 256 * wgs_per_cu;   // 4 wavefronts of a workgroup on 4 SIMDs
                printf("{\"taps_ahead\": %d, \"table_kb\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_fmac_per_simd\": %.3f}\n", ahead, kb, wgs_per_cu, ms, ms * 1e6 / fmacs_per_simd);
            }
    return 0;
}
