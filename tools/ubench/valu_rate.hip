// VALU / transcendental issue-rate micro-benchmark for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Prints cycles per wave-instruction per SIMD for a few instruction mixes at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 0.999f, d = 1e-3f;
    f32x2 c2 = {c, c}, d2 = {d, d};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));)
        } else if (MODE == 2) {  // 8 independent v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {  // 4 exp interleaved with 4 pk_fma (co-issue?)
            REP8(asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_pk_fma_f32 %5, %5, %8, %9\n"
                         "v_exp_f32 %2, %2\n v_pk_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2), "v"(d2));)
        } else if (MODE == 4) {  // 4 exp interleaved with 4 v_fma
            REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32 %1, %1\n v_fma_f32 %5, %5, %8, %9\n"
                         "v_exp_f32 %2, %2\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %3, %3\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
        } else if (MODE == 5) {  // 8 v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 6) {  // 8 v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c2));)
        } else if (MODE == 7) {  // 8 v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 8) {  // 8 v_exp_f16 (fp16 transcendental)
            REP8(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n"
                         "v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const char* name, float* d_out) {
    const int iters = 4000;
    for (int wps : {1, 2, 4, 8}) {
        int nblocks = 256 * 4 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(64), 0, 0, d_out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(nblocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double inst_per_simd = (double)iters * 64 * wps;
        double ns_per_inst = ms * 1e6 / inst_per_simd;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns/wave-inst/SIMD  = %.2f cyc @2.4GHz\n", name, wps, ms, ns_per_inst, ns_per_inst * 2.4);
    }
}

int main() {
    float* d_out; hipMalloc(&d_out, 4);
    run<0>("v_fma_f32", d_out);
    run<1>("v_pk_fma_f32", d_out);
    run<5>("v_mul_f32", d_out);
    run<6>("v_pk_mul_f32", d_out);
    run<2>("v_exp_f32", d_out);
    run<7>("v_rcp_f32", d_out);
    run<8>("v_exp_f16", d_out);
    run<3>("exp+pk_fma 1:1", d_out);
    run<4>("exp+fma 1:1", d_out);
    return 0;
}
