// Micro-benchmarks that size the mel kernel's floors on gfx950 (diagnostics, not product code):
// issue cost of packed / scalar fp32 VALU, DPP moves, v_log_f32 and of the LDS access patterns the kernel uses,
// at 1..4 waves per SIMD.      hipcc --offload-arch=gfx950 -O3 -o valu_lds valu_lds.hip && ./valu_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 256;

#define REP8(x) x x x x x x x x
#define REP16(x) REP8(x) REP8(x)

template <int OP>
__global__ __launch_bounds__(1024) void valu_kernel(unsigned long long *out, float seed) {
    f2 a[8], b = {seed, seed * 0.5f}, c = {0.25f, 0.125f};
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i, seed - i};
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = seed + i;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
        if (OP == 0) {   // v_fma_f32, 8 independent chains x 2
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(b.x), "v"(c.x));
        } else if (OP == 1) {   // v_pk_fma_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (OP == 2) {   // v_pk_add_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (OP == 3) {   // v_pk_mul_f32 with op_sel broadcast
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a[i]) : "v"(b));
        } else if (OP == 4) {   // DPP row_mirror move
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(s[i]));
        } else if (OP == 5) {   // v_log_f32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(s[i]));
        } else if (OP == 6) {   // dependent v_pk_fma chain (latency)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
        } else if (OP == 7) {   // v_add_f32 with DPP source (row_mirror) : a fused exchange + add
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(b.x));
        } else if (OP == 8) {   // v_cndmask_b32
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b.x));
        }
    }
    const unsigned long long t1 = clock64();
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y + s[i];
    if (acc == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// LDS patterns.  PAT 0: ds_read_b64 transpose gather (lane*17 + n2 f2 units, 4 regions of 544 floats per wave)
//                PAT 1: ds_write_b64 transpose scatter (k1*17 + lane)
//                PAT 2: ds_read2_b32 sample pairs offsets (0,160): lanes 2l, frames 320 apart per 16-lane group (the 2-way conflict)
//                PAT 3: ds_read2_b32 on de-interleaved samples: lane l, groups 80 apart (conflict-free variant)
//                PAT 4: ds_read_b128 window rows (q*16 + l float4, all groups same address)
//                PAT 5: ds_write_b128 staging (tid*4 floats)
//                PAT 6: 2 x ds_write_b64 staging (de-interleaved)
//                PAT 7: ds_read_b32 filterbank weights (slot*16 + l), PAT 8: ds_read_b64 bins gather P2[lo + j] with lo = 2*l (low mels) 
template <int PAT>
__global__ __launch_bounds__(1024) void lds_kernel(unsigned long long *out, int zero) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, l = lane & 15, g = lane >> 4, w = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    float *base = lds + (w & 3) * 4096 + zero;   // each wave its own 16 KB window (waves 4.. share with 0..3: reads only matter)
    f2 v[16];
    float4 q4[8];
    for (int i = 0; i < 16; ++i) v[i] = f2{1.0f * i, 2.0f * i};
    float acc = 0;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
        if (PAT == 0) {
            const f2 *r = reinterpret_cast<const f2 *>(base + g * 544) + l * 17;
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = r[n];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(v[n]));
        } else if (PAT == 1) {
            f2 *r = reinterpret_cast<f2 *>(base + g * 544) + l;
#pragma unroll
            for (int n = 0; n < 16; ++n) r[n * 17] = v[n];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (PAT == 2) {
            const float *x = base + g * 320 + 2 * l;
#pragma unroll
            for (int n = 0; n < 8; ++n) { v[2 * n] = f2{x[32 * n], x[32 * n + 160]}; v[2 * n + 1] = f2{x[32 * n + 1], x[32 * n + 161]}; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(v[n]));
        } else if (PAT == 3) {
            const float *x = base + (g >> 1) * 320 + (g & 1) * 80 + l;   // groups of a 32-lane half 80 floats apart: banks +16
#pragma unroll
            for (int n = 0; n < 8; ++n) { v[2 * n] = f2{x[16 * n], x[16 * n + 160]}; v[2 * n + 1] = f2{x[16 * n + 2048], x[16 * n + 2048 + 160]}; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(v[n]));
        } else if (PAT == 4) {
#pragma unroll
            for (int n = 0; n < 8; ++n) q4[n] = reinterpret_cast<const float4 *>(base)[n * 16 + l];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 8; ++n) asm volatile("" : "+v"(q4[n].x), "+v"(q4[n].y), "+v"(q4[n].z), "+v"(q4[n].w));
        } else if (PAT == 5) {
#pragma unroll
            for (int n = 0; n < 4; ++n) reinterpret_cast<float4 *>(base)[n * 64 + lane] = float4{v[n].x, v[n].y, v[n + 1].x, v[n + 1].y};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (PAT == 6) {
#pragma unroll
            for (int n = 0; n < 4; ++n) { reinterpret_cast<f2 *>(base)[n * 64 + lane] = v[n]; reinterpret_cast<f2 *>(base + 2048)[n * 64 + lane] = v[n + 1]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (PAT == 7) {
            float t[16];
#pragma unroll
            for (int n = 0; n < 16; ++n) t[n] = base[n * 16 + l];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(t[n]));
        } else if (PAT == 8) {
            const f2 *r = reinterpret_cast<const f2 *>(base + g * 544) + 2 * l;
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = r[n & 1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 16; ++n) asm volatile("" : "+v"(v[n]));
        }
    }
    const unsigned long long t1 = clock64();
    for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
    for (int i = 0; i < 8; ++i) acc += q4[i].x;
    if (acc == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <class K, class... A>
double run(K kern, int threads, size_t lds, A... args) {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 16);
    hipMemset(d, 0, 16);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d, args...);   // one workgroup per CU, `threads`/64 waves on it
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), lds, 0, d, args...);
    hipDeviceSynchronize();
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return static_cast<double>(h);
}

int main() {
    const char *vn[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32 op_sel", "v_mov_b32_dpp row_mirror", "v_log_f32", "v_pk_fma_f32 dependent chain", "v_add_f32_dpp", "v_cndmask_b32"};
    printf("VALU: cycles per wave-instruction as seen by ONE wave (x waves/SIMD = SIMD issue cost when throughput-bound)\n");
    for (int op = 0; op < 9; ++op) {
        printf("%-32s", vn[op]);
        for (int wps = 1; wps <= 4; ++wps) {
            const int threads = 256 * wps;
            double c = 0;
            switch (op) {
                case 0: c = run(valu_kernel<0>, threads, 0, 1.0f); break;
                case 1: c = run(valu_kernel<1>, threads, 0, 1.0f); break;
                case 2: c = run(valu_kernel<2>, threads, 0, 1.0f); break;
                case 3: c = run(valu_kernel<3>, threads, 0, 1.0f); break;
                case 4: c = run(valu_kernel<4>, threads, 0, 1.0f); break;
                case 5: c = run(valu_kernel<5>, threads, 0, 1.0f); break;
                case 6: c = run(valu_kernel<6>, threads, 0, 1.0f); break;
                case 7: c = run(valu_kernel<7>, threads, 0, 1.0f); break;
                case 8: c = run(valu_kernel<8>, threads, 0, 1.0f); break;
            }
            printf("  %dw/SIMD: %6.2f (per SIMD %5.2f)", wps, c / (kIters * 16.0), c / (kIters * 16.0) / wps);
        }
        printf("\n");
    }
    const char *pn[] = {"ds_read_b64 transpose gather", "ds_write_b64 transpose scatter", "ds_read2_b32 samples (2-way)", "ds_read2_b32 samples de-interleaved", "ds_read_b128 window rows",
                        "ds_write_b128 staging", "2x ds_write_b64 staging", "ds_read_b32 weights", "ds_read_b64 bins (low mels)"};
    const int per_iter[] = {16, 16, 16, 16, 8, 4, 8, 16, 16};
    printf("LDS: cycles per wave-instruction seen by one wave; (per CU) = that / waves on the CU = LDS-pipe cost when throughput-bound\n");
    for (int p = 0; p < 9; ++p) {
        printf("%-36s", pn[p]);
        for (int waves = 4; waves <= 16; waves *= 2) {
            double c = 0;
            const int threads = 64 * waves;
            switch (p) {
                case 0: c = run(lds_kernel<0>, threads, 65536, 0); break;
                case 1: c = run(lds_kernel<1>, threads, 65536, 0); break;
                case 2: c = run(lds_kernel<2>, threads, 65536, 0); break;
                case 3: c = run(lds_kernel<3>, threads, 65536, 0); break;
                case 4: c = run(lds_kernel<4>, threads, 65536, 0); break;
                case 5: c = run(lds_kernel<5>, threads, 65536, 0); break;
                case 6: c = run(lds_kernel<6>, threads, 65536, 0); break;
                case 7: c = run(lds_kernel<7>, threads, 65536, 0); break;
                case 8: c = run(lds_kernel<8>, threads, 65536, 0); break;
            }
            printf("  %2d waves/CU: %6.2f (per CU %5.2f)", waves, c / (kIters * (double)per_iter[p]), c / (kIters * (double)per_iter[p]) / waves);
        }
        printf("\n");
    }
    return 0;
}
