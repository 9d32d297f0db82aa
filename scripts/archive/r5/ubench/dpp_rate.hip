// Round 5 microbenchmark: issue rate of v_fmac_f32_dpp row_newbcast (the tap broadcast of the resampler's row kernels) against plain v_fmac_f32, by number of
// independent accumulator chains and by wavefronts per SIMD.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 dpp_rate.hip -o dpp_rate && ./dpp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE, int CHAINS>
__global__ void k(float *out, int iters) {
    float acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = threadIdx.x * 1e-9f + c;
    float t = threadIdx.x * 1e-3f, x0 = 1.0001f, x1 = 0.9999f;
    asm volatile("" : "+v"(t), "+v"(x0), "+v"(x1));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if constexpr (MODE == 0) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc[c]) : "v"(t), "v"(r & 1 ? x0 : x1));
                else if constexpr (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[c]) : "v"(t), "v"(r & 1 ? x0 : x1));
                else asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc[c]) : "v"(t), "v"(r & 1 ? x0 : x1));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int CHAINS>
void run(const char *name, float *d, int waves_per_simd) {
    const int iters = 2000, threads = 256 * waves_per_simd, grid = 256;     // one workgroup per CU (by size: nothing else keeps two off one CU, so read the rate, not the total)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(grid), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(grid), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_wave = double(iters) * 32 * CHAINS;
    printf("{\"op\": \"%s\", \"chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_instruction_per_wave\": %.3f, \"ns_per_instruction_per_simd\": %.3f}\n", name, CHAINS, waves_per_simd, ms,
           ms * 1e6 / inst_per_wave, ms * 1e6 / (inst_per_wave * waves_per_simd));
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 1024 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0, 1>("fmac_dpp_row_newbcast", d, w); run<0, 2>("fmac_dpp_row_newbcast", d, w); run<0, 4>("fmac_dpp_row_newbcast", d, w); run<0, 8>("fmac_dpp_row_newbcast", d, w);
        run<1, 1>("fmac", d, w); run<1, 2>("fmac", d, w); run<1, 4>("fmac", d, w); run<1, 8>("fmac", d, w);
        run<2, 2>("fmac_dpp_quad_perm", d, w); run<2, 4>("fmac_dpp_quad_perm", d, w);
    }
    return 0;
}
