// Can a persistent kernel whose workgroups all sit on ONE XCD exchange per-round records through that XCD's L2 (no kernel
// boundary, no device-scope fences)?  Measures: XCC_ID of every workgroup (round-robin?), the join protocol, the latency of
// one round = publish a stamped 16-byte record, poll until all G records carry the round's stamp, one dependent HBM load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Ctl { unsigned arrived, joined, pad[30]; unsigned xcc[512]; unsigned long long cycles[256]; unsigned long long sum[256]; };

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }

template <int T>
__global__ __launch_bounds__(T) void k(Ctl *ctl, u4 *rec /*[2][64]*/, const double *M, long long Np, int rounds, int maxG, int loads, int rowmode, int anyxcc, int elems) {
    __shared__ int s_ticket, s_G;
    __shared__ unsigned long long s_row;
    extern __shared__ double dyn[];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        const unsigned x = xcc_id();
        ctl->xcc[blockIdx.x] = x;
        int ticket = -1;
        if (x == 0 || anyxcc) { ticket = static_cast<int>(atomicAdd(&ctl->joined, 1u)); if (ticket >= maxG) ticket = -1; }
        __threadfence();
        atomicAdd(&ctl->arrived, 1u);
        if (ticket >= 0) {
            for (long long spin = 0; __hip_atomic_load(&ctl->arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spin < 20000000; ++spin) __builtin_amdgcn_s_sleep(2);
            const unsigned j = __hip_atomic_load(&ctl->joined, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            s_G = static_cast<int>(j) < maxG ? static_cast<int>(j) : maxG;
        }
        s_ticket = ticket;
    }
    __syncthreads();
    const int g = s_ticket;
    if (g < 0) return;
    const int G = s_G;
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(rec, 0, 2 * 256 * 8 * 16, 0x00027000);
    const long long t0 = clock64();
    double acc = 0.0;
    unsigned long long row = 12345 + g;
    for (int t = 1; t <= rounds; ++t) {
        const int par = t & 1;
        // poll: lane l < G reads record l of this parity until every stamp equals t (round 1: published below first)
        if (t > 1) {
            bool timeout = false;
            if (tid < 64 || !anyxcc) {   // anyxcc: wave 0 polls for the workgroup (4 records per lane), the others wait at the barrier
                for (long long spin = 0;; ++spin) {
                    if (spin > 5000000) { timeout = true; break; }   // never hang the box
                    bool bad = false;
                    unsigned sx = 0;
                    u4 v[4][7];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = lane * 4 + j;
                        const int ii = i < G ? i : 0;   // every load is issued before anything is compared
#pragma unroll
                        for (int e = 0; e < 7; ++e)
                            if (e < elems) v[j][e] = __builtin_amdgcn_raw_buffer_load_b128(rr, ((par * 256 + ii) * 8 + e) * 16, 0, 16 /* sc1 */);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 7; ++e)
                            if (e < elems) { bad = bad | (v[j][e].w != static_cast<unsigned>(t)); sx += v[j][e].x + v[j][e].y; }
                    if (__builtin_amdgcn_ballot_w64(bad) == 0ull) { row = row * 6364136223846793005ull + __builtin_amdgcn_readfirstlane(sx); break; }
                }
            }
            if (anyxcc) {
                if (tid == 0) { s_row = timeout ? ~0ull : row; }
                __syncthreads();
                row = s_row;
                timeout = row == ~0ull;
                __syncthreads();
            }
            if (timeout) { if (tid == 0) ctl->pad[0] = t; return; }
        }
        // dependent HBM loads: the row index comes out of the exchange
        // rowmode 0: random row (as the merge loop), 1: the same row every round, 2: consecutive rows
        const unsigned long long r = rowmode == 0 ? ((row >> 33) * 40000ull) >> 31 : (rowmode == 1 ? 777ull : static_cast<unsigned long long>(t));
        const double *src = M + r * Np + (g * loads) * T + tid;
        for (int c = 0; c < loads; ++c) acc += __builtin_nontemporal_load(src + c * T);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const u4 o = {static_cast<unsigned>(t * 3 + g), static_cast<unsigned>(acc), 0u, static_cast<unsigned>(t + 1)};
            for (int e = 0; e < elems; ++e) __builtin_amdgcn_raw_buffer_store_b128(o, rr, ((((t + 1) & 1) * 256 + g) * 8 + e) * 16, 0, 16);
        }
    }
    if (tid == 0) { ctl->cycles[g] = clock64() - t0; ctl->sum[g] = static_cast<unsigned long long>(acc) + row; }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    Ctl *ctl; u4 *rec; double *M;
    const long long Np = 50176;
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&rec, 2 * 256 * 8 * 16)); CK(hipMalloc(&M, Np * Np * 8));
    CK(hipMemset(M, 0, Np * Np * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // {G, loads per thread, row mode, any XCD, 16-byte elements per record}
    const int cfgs[][5] = {{32, 0, 0, 0, 1}, {32, 0, 0, 0, 7}, {32, 1, 0, 0, 7}, {196, 0, 0, 1, 1}, {196, 0, 0, 1, 7}, {196, 1, 0, 1, 7}, {196, 2, 0, 1, 7}, {196, 5, 0, 1, 7}, {256, 0, 0, 1, 7}, {64, 0, 0, 1, 7}, {64, 3, 0, 1, 7}};
    for (int cfg = 0; cfg < 11; ++cfg) {
        const int maxG = cfgs[cfg][0], loads = cfgs[cfg][1], rowmode = cfgs[cfg][2], anyxcc = cfgs[cfg][3], elems = cfgs[cfg][4];
        CK(hipMemset(ctl, 0, sizeof(Ctl))); CK(hipMemset(rec, 0, 2 * 256 * 8 * 16));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<512>, dim3(256), dim3(512), 90 * 1024, 0, ctl, rec, M, Np, rounds, maxG, loads, rowmode, anyxcc, elems);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        if (cfg == 0) { printf("xcc of workgroups 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", h.xcc[i]); printf("\n"); int bad = 0; for (int i = 0; i < 256; ++i) bad += h.xcc[i] != static_cast<unsigned>(i & 7); printf("workgroups off the round-robin XCD: %d of 256\n", bad); }
        if (h.pad[0]) printf("TIMEOUT in round %u\n", h.pad[0]);
        printf("anyxcc=%d elems=%d G=%2u joined=%u rowmode=%d loads/thread=%d rounds=%d: %.3f ms = %.3f us per round (%.0f cycles, %.1f KB of rows per round)\n", anyxcc, elems, maxG < (int)h.joined ? maxG : h.joined, h.joined, rowmode, loads, rounds, ms, 1e3 * ms / rounds,
               static_cast<double>(h.cycles[0]) / rounds, loads * 512.0 * (maxG < (int)h.joined ? maxG : h.joined) * 8 / 1024);
    }
    return 0;
}
