// Round-5 follow-up of pk_opsel.hip (which replayed the HIGH-dword selection on SRC0 only, the slot the K4x bisect found clean):
// the selection on SRC1 -- v_pk_mul_f32 d, g, p op_sel:[0,1] op_sel_hi:[1,1], the form that fails in conv_xproj_bwd_slab_kernel
// (profiles/r05_k4x_repro.txt) -- in isolation, next to a stream of MFMAs of the same wave, and next to MFMAs + LDS traffic.
//   D: v_pk_mul_f32 d, g, p op_sel:[0,1] op_sel_hi:[1,1]     p = {junk, 1.0f} from LDS (one address per wave)
//   E: v_pk_mul_f32 d, p, g op_sel:[1,0] op_sel_hi:[1,1]     the same value through src0 (clean in the kernel)
// hipcc --offload-arch=gfx950 -O2 pk_opsel_src1.hip -o pk_opsel_src1 && ./pk_opsel_src1
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// MODE 0: no MFMA, 1: two MFMAs per iteration in front of the multiplies, 2: MFMAs + their results through LDS, 3: the LDS traffic of 2
// without the MFMAs.  VAR (with the traffic of MODE): 0 D then E, 1 E then D, 2 / 5 / 6 / 7: s_nop 7 x 4 / 1 / 2 / 3 in front of D, 3 s_waitcnt lgkmcnt(0) in front
// of D, 4 the pair p made by VALU (v_mov from the LDS value a full iteration earlier) instead of read from LDS right before
template <int MODE, int VAR = 0>
__global__ __launch_bounds__(512, 2) void probe(unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(16))) f2 tab_hi[64];
    __shared__ __attribute__((aligned(16))) float pt[8][16 * 132];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) tab_hi[tid] = (f2){(float)(tid * 1056), 1.0f};      // low dword: a small integer's bits would be a denormal; any junk will do
    __syncthreads();
    unsigned nD = 0, nE = 0;
    f2 g = {1.0f + 0.001f * tid, 2.0f + 0.003f * tid};
    u4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u + lane, 0x3f803f80u}, b = {0x3f803f80u, 0x40004000u, 0x3f803f80u, 0x3f803f80u + lane};
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    float sink = 0.f;
    f2 p_prev = tab_hi[wave & 63];
    for (int it = 0; it < iters; ++it) {
        f2 p = tab_hi[(it + wave) & 63];
        if (VAR == 4) { const f2 t = p; p = p_prev; p_prev = t; asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1" : "+v"(p.x), "+v"(p.y)); }
        if (MODE == 1 || MODE == 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, b), __builtin_bit_cast(bf8, a), acc, 0, 0, 0);
        }
        if (MODE >= 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[wave][(4 * (lane >> 4) + r) * 132 + (lane & 15)] = acc[r];
            const f2 pv = *reinterpret_cast<const f2*>(&pt[wave][(it & 15) * 132 + 2 * lane % 128]);
            sink += pv.x;
        }
        const float e = __builtin_amdgcn_exp2f(-g.x * 0.01f), rc = __builtin_amdgcn_rcpf(1.0f + e);
        sink += rc;
        f2 dD, dE;
        if (VAR == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(dE) : "v"(p), "v"(g));
        if (VAR == 2) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        else if (VAR == 5) asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        else if (VAR == 6) asm volatile("s_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        else if (VAR == 7) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        else if (VAR == 3) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(dD) : "v"(g), "v"(p));
        if (VAR != 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(dE) : "v"(p), "v"(g));
        nD += (dD.x != g.x) | (dD.y != g.y);
        nE += (dE.x != g.x) | (dE.y != g.y);
        g.x += 0.5f; g.y += 0.25f;
        if (g.x > 1000.f) { g.x -= 999.f; g.y -= 499.f; }
        if ((MODE == 1 || MODE == 2) && (it & 255) == 255) acc = (f4){0.f, 0.f, 0.f, 0.f};
    }
    if (sink + acc[0] == 12345.678f) nD += 1000000;
    atomicAdd(&bad[0 * 4 + (lane >> 4)], nD);
    atomicAdd(&bad[1 * 4 + (lane >> 4)], nE);
}

template <int MODE, int VAR = 0> static void run(const char* what) {
    unsigned* d; unsigned h[8] = {0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int iters = 20000, wgs = 2048;
    hipLaunchKernelGGL((probe<MODE, VAR>), dim3(wgs), dim3(512), 0, 0, d, iters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s: %s; %d workgroups x 512 lanes x %d products per form; mismatches by 16-lane group [0-15, 16-31, 32-47, 48-63]\n", what, hipGetErrorString(e), wgs, iters);
    printf("   D src1 high (op_sel:[0,1])  %u %u %u %u\n   E src0 high (op_sel:[1,0])  %u %u %u %u\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    hipFree(d);
}

int main() {
    run<0>("no MFMA");
    run<1>("2 MFMAs per iteration");
    run<2>("2 MFMAs + product tile through LDS per iteration");
    run<3>("the LDS traffic alone (4 ds_write_b32 + ds_read_b64 per iteration, no MFMA)");
    run<2, 1>("MFMAs + LDS, E issued BEFORE D");
    run<2, 2>("MFMAs + LDS, 4 x s_nop 7 in front of D");
    run<2, 5>("MFMAs + LDS, 1 x s_nop 7 in front of D");
    run<2, 6>("MFMAs + LDS, 2 x s_nop 7 in front of D");
    run<2, 7>("MFMAs + LDS, 3 x s_nop 7 in front of D");
    run<2, 3>("MFMAs + LDS, s_waitcnt vmcnt(0) lgkmcnt(0) in front of D");
    run<2, 4>("MFMAs + LDS, the pair read from LDS a whole iteration earlier and passed through v_mov");
    return 0;
}
