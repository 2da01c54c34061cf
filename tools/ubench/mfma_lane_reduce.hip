// Checks the lane->element maps used by the MFMA cross-lane reduction of scan_bwd (K2): with 0/1 selector A fragments,
// D[4*(lane>>4)+r][lane&15] = sum over the 4 lane groups of register (4*(lane>>4)+r) of lane (lane&15)+16g.
// hipcc --offload-arch=gfx950 -O3 -o build/mfma_lane_reduce mfma_lane_reduce.hip
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* in) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x * 16 + i];
    const int lane = threadIdx.x & 63;
    u32x4 a1, a2, b1, b2;
    for (int p = 0; p < 4; ++p) {
        unsigned lo1 = ((lane & 15) == 2 * p) ? 0x3F80u : 0u, hi1 = ((lane & 15) == 2 * p + 1) ? 0x3F80u : 0u;
        unsigned lo2 = ((lane & 15) == 2 * p + 8) ? 0x3F80u : 0u, hi2 = ((lane & 15) == 2 * p + 9) ? 0x3F80u : 0u;
        a1[p] = lo1 | (hi1 << 16);
        a2[p] = lo2 | (hi2 << 16);
        unsigned x, y;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(v[2 * p]), "v"(v[2 * p + 1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(v[8 + 2 * p]), "v"(v[8 + 2 * p + 1]));
        b1[p] = x; b2[p] = y;
    }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a2), __builtin_bit_cast(bf16x8, b2), d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[threadIdx.x * 4 + r] = d[r];
}
int main() {
    float *in, *out;
    (void)hipHostMalloc(&in, 64 * 16 * 4); (void)hipHostMalloc(&out, 64 * 4 * 4);
    for (int l = 0; l < 64; ++l) for (int m = 0; m < 16; ++m) in[l * 16 + m] = (float)((l * 7 + m * 3) % 13 - 6);   // exact in bf16
    k<<<1, 64>>>(out, in);
    (void)hipDeviceSynchronize();
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int m = 4 * (l >> 4) + r, j = l & 15;
        float ref = 0; for (int g = 0; g < 4; ++g) ref += in[(j + 16 * g) * 16 + m];
        if (ref != out[l * 4 + r]) { if (bad < 8) printf("lane %d r %d got %f want %f\n", l, r, out[l * 4 + r], ref); ++bad; }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
