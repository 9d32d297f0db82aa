// LDS bytes per clock and CU by instruction width (diagnostics): does a 16-byte write (ds_write2_b64 / ds_write_b128) move more bytes per
// clock than two 8-byte writes?  12 wavefronts per CU (the mel kernel's occupancy), conflict-free addresses, hipEvents.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 2048;
// OP 0 ds_write_b32 | 1 ds_write_b64 | 2 ds_write2_b64 | 3 ds_write_b128 | 4 ds_read_b64 | 5 ds_read2_b64 | 6 ds_read_b128 | 7 ds_read2_b32 | 8 ds_read_b32
template <int OP>
__global__ __launch_bounds__(768) void lds(float *out) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += blockDim.x) sm[i] = i;
    __syncthreads();
    const unsigned a4 = tid * 4, a8 = tid * 8, a16 = tid * 16;   // byte addresses: consecutive lanes, consecutive elements
    float s = tid;
    f2 v = {1.0f * tid, 2.0f}, u = {3.0f, 4.0f};
    f4 q = {1.0f, 2.0f, 3.0f, 1.0f * tid};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (OP == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a4), "v"(s), "n"(n * 3072) : "memory");
            if (OP == 1) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(v), "n"((n & 3) * 6144) : "memory");
            if (OP == 2) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a8), "v"(v), "v"(u), "n"(0), "n"(192) : "memory");   // 8-byte units: +768 elements would not fit offset1; two rows 1536 B apart
            if (OP == 3) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(q), "n"((n & 1) * 12288) : "memory");
            if (OP == 4) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a8), "n"((n & 3) * 6144) : "memory");
            if (OP == 5) { f4 r; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(a8), "n"(0), "n"(192) : "memory"); q = r; }
            if (OP == 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(a16), "n"((n & 1) * 12288) : "memory");
            if (OP == 7) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a4), "n"(0), "n"(192) : "memory");
            if (OP == 8) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s) : "v"(a4), "n"(n * 3072) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float acc = s + v.x + v.y + q.x + q.w;
    if (acc == 12345.678f) out[0] = acc;
}
template <class K>
float run(K k, int blocks, int threads, size_t ldsb) {
    float *d;
    (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), ldsb, 0, d);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), ldsb, 0, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}

// Patterns of the mel kernel (lane stride in bytes = run-time argument, offsets = template arguments, in units of the access width):
// OP 0 ds_read_b64 x2 (offsets O0, O1) | 1 ds_read2_b64 (O0, O1) | 2 ds_read_b128 (O0 in 16-byte units) | 3 ds_write_b64 x2 | 4 ds_write2_b64 | 5 ds_read2_b32 | 6 ds_read_b32 x2
template <int OP, int O0, int O1>
__global__ __launch_bounds__(768) void pat(float *out, int lane_stride, int group_stride) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += blockDim.x) sm[i] = i;
    __syncthreads();
    const unsigned a = ((tid & 15) * lane_stride + (tid >> 4) * group_stride) % 40960;   // 16-lane groups as in the kernel
    f2 v = {1.0f * tid, 2.0f}, u = {3.0f, 4.0f};
    f4 q = {1.0f, 2.0f, 3.0f, 1.0f * tid};
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (OP == 0) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(O0 * 8) : "memory"); asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(u) : "v"(a), "n"(O1 * 8) : "memory"); }
            if (OP == 1) { f4 r; asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r) : "v"(a), "n"(O0), "n"(O1) : "memory"); q = r; }
            if (OP == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(a), "n"(O0 * 16) : "memory");
            if (OP == 3) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(O0 * 8) : "memory"); asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a), "v"(u), "n"(O1 * 8) : "memory"); }
            if (OP == 4) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a), "v"(v), "v"(u), "n"(O0), "n"(O1) : "memory");
            if (OP == 5) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(a), "n"(O0), "n"(O1) : "memory");
            if (OP == 6) { float s0, s1; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s0) : "v"(a), "n"(O0 * 4) : "memory"); asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s1) : "v"(a), "n"(O1 * 4) : "memory"); v.x = s0; v.y = s1; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float acc = v.x + v.y + q.x + q.w + u.x;
    if (acc == 12345.678f) out[0] = acc;
}
template <class K>
float run2(K k, int ls, int gs) {
    float *d;
    (void)hipMalloc(&d, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 49152, 0, d, ls, gs);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(768), 49152, 0, d, ls, gs);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms;
}
void report(const char *what, float ms, int bytes_per_lane_per_step) {
    const double steps = 12.0 * kIters * 8.0, clk = ms * 1e-3 * 2.4e9;
    printf("%-78s %6.3f ms: %5.2f clk per 64 lanes x %2d B, %6.1f B/clk/CU\n", what, ms, clk / steps, bytes_per_lane_per_step, steps * 64.0 * bytes_per_lane_per_step / clk);
}
void patterns() {
    // region stride of the kernel: 544 floats = 2176 B (17-pair rows) or 2304 B (18-pair rows)
    report("transpose get, rows of 17 pairs: ds_read2_b64 adjacent, lane stride 136 B", run2(pat<1, 0, 1>, 136, 2176), 16);
    report("transpose get, rows of 17 pairs: 2 x ds_read_b64, lane stride 136 B", run2(pat<0, 0, 1>, 136, 2176), 16);
    report("transpose get, rows of 18 pairs: ds_read_b128, lane stride 144 B", run2(pat<2, 0, 0>, 144, 2304), 16);
    report("transpose get, rows of 18 pairs: ds_read2_b64 adjacent, lane stride 144 B", run2(pat<1, 0, 1>, 144, 2304), 16);
    report("transpose put: 2 x ds_write_b64, rows 136 B apart, lane stride 8 B", run2(pat<3, 0, 17>, 8, 2176), 16);
    report("transpose put: ds_write2_b64, rows 136 B apart, lane stride 8 B", run2(pat<4, 0, 17>, 8, 2176), 16);
    report("transpose put: 2 x ds_write_b64, rows 144 B apart, lane stride 8 B", run2(pat<3, 0, 18>, 8, 2304), 16);
    report("transpose put: ds_write2_b64, rows 144 B apart, lane stride 8 B", run2(pat<4, 0, 18>, 8, 2304), 16);
    report("twiddle table [k][lane]: ds_read2_b64 rows 128 B apart, lane stride 8 B, all groups same", run2(pat<1, 0, 16>, 8, 0), 16);
    report("twiddle table [k][lane]: 2 x ds_read_b64 rows 128 B apart, lane stride 8 B, all groups same", run2(pat<0, 0, 16>, 8, 0), 16);
    report("twiddle table [lane][k]: ds_read_b128, lane stride 144 B, all groups same", run2(pat<2, 0, 0>, 144, 0), 16);
    report("samples: ds_read2_b32 offsets 16 / 176, lane stride 4 B, groups 80 floats apart", run2(pat<5, 16, 176>, 4, 320), 8);
    report("samples: 2 x ds_read_b32 offsets 16 / 176, lane stride 4 B, groups 80 floats apart", run2(pat<6, 16, 176>, 4, 320), 8);
    report("power pairs: ds_read2_b64 adjacent, lane stride 8 B (neighbours overlap)", run2(pat<1, 0, 1>, 8, 2176), 16);
    report("power pairs: 2 x ds_read_b64 adjacent, lane stride 8 B", run2(pat<0, 0, 1>, 8, 2176), 16);
    report("power pairs: ds_read2_b64 adjacent, lane stride 16 B", run2(pat<1, 0, 1>, 16, 2176), 16);
    report("power pairs: ds_read2_b64 adjacent, lane stride 24 B", run2(pat<1, 0, 1>, 24, 2176), 16);
}
int main() {
    const char *name[9] = {"ds_write_b32", "ds_write_b64", "ds_write2_b64", "ds_write_b128", "ds_read_b64", "ds_read2_b64", "ds_read_b128", "ds_read2_b32", "ds_read_b32"};
    const int width[9] = {4, 8, 16, 16, 8, 16, 16, 8, 4};
    float ms[9];
    for (int rep = 0; rep < 2; ++rep) {
        ms[0] = run(lds<0>, 256, 768, 49152); ms[1] = run(lds<1>, 256, 768, 49152); ms[2] = run(lds<2>, 256, 768, 49152);
        ms[3] = run(lds<3>, 256, 768, 49152); ms[4] = run(lds<4>, 256, 768, 49152); ms[5] = run(lds<5>, 256, 768, 49152);
        ms[6] = run(lds<6>, 256, 768, 49152); ms[7] = run(lds<7>, 256, 768, 49152); ms[8] = run(lds<8>, 256, 768, 49152);
    }
    for (int op = 0; op < 9; ++op) {
        const double instr = 12.0 * kIters * 8.0;   // wave-instructions per CU
        const double clk = ms[op] * 1e-3 * 2.4e9;
        printf("%-14s %6.3f ms: %5.2f clk per wave-instruction and CU, %6.1f B/clk/CU (2.4 GHz assumed)\n", name[op], ms[op], clk / instr, instr * 64.0 * width[op] / clk);
    }
    patterns();
    return 0;
}
