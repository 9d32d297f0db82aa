// Raw-buffer range checking on gfx950: is a dwordx4 load that straddles num_records checked per dword? negative offsets? soffset?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *x, int nrec_bytes, float *out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, nrec_bytes, 0x00027fac);
    const int t = threadIdx.x;   // lane t loads 16 bytes at byte offset 4 * (t - 4): lanes 0..3 start below 0
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 4 * (t - 4), 0, 0);
    const u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, 4 * t - 1024, 1008, 0);   // the same addresses through soffset
    u4 *o4 = reinterpret_cast<u4 *>(out);
    o4[2 * t] = v;
    o4[2 * t + 1] = w;
}
int main() {
    float h[64], *d, *o, ho[64 * 8];
    for (int i = 0; i < 64; ++i) h[i] = 100 + i;
    hipMalloc(&d, 4096); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int base_off = 0; base_off < 2; ++base_off) {   // aligned and 4-byte-misaligned base
        const int nrec = 4 * 22;                          // 22 valid floats
        hipLaunchKernelGGL(k, dim3(1), dim3(32), 0, 0, d + base_off, nrec, o);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("base +%d floats, num_records = 22 floats; lane: voffset-only load | soffset load\n", base_off);
        for (int t = 0; t < 32; ++t) printf("  lane %2d (first float index %3d): %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f %6.0f %6.0f\n", t, t - 4, ho[8 * t], ho[8 * t + 1], ho[8 * t + 2],
                                            ho[8 * t + 3], ho[8 * t + 4], ho[8 * t + 5], ho[8 * t + 6], ho[8 * t + 7]);
    }
    return 0;
}
