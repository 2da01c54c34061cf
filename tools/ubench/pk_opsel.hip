// Does a packed-fp32 multiply that takes its LOW result's operand from the HIGH dword of a register pair (op_sel:[1,0]) -- or from a
// v_mov_b32 issued right in front of it -- always see the right value on gfx950?  Background: DESIGN.md section 3 (K4x slab form):
// with the row-table entry laid out {acc_off, own} hipcc emitted exactly these two forms for `float2 * own`, and the kernel lost
// single rows of dw / db in lanes 48-63, first channel, now and then.  This probe replays the forms in isolation: every lane reads
// the pair {junk, 1.0f} from LDS (one address for the whole wave, as the row table is read), multiplies its own float2 by the
// broadcast 1.0f in three ways, and counts results that are not bit-identical to its float2.
//   A: v_pk_mul_f32 d, p, g op_sel:[1,0]                       (low result from p.hi)
//   B: v_mov_b32 t.lo, p.hi ; v_pk_mul_f32 d, t, g op_sel_hi:[0,1]   (copy, then the ordinary low-dword broadcast)
//   C: v_pk_mul_f32 d, q, g op_sel_hi:[0,1] with q = {1.0f, junk} read from LDS   (the form every kernel here uses)
// hipcc --offload-arch=gfx950 -O2 pk_opsel.hip -o pk_opsel && ./pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512, 1) void probe(unsigned* bad, int iters) {
    __shared__ __attribute__((aligned(8))) f2 tab_hi[64], tab_lo[64];
    __shared__ float big[36 * 1024];                                   // 144 KB: one workgroup per CU, like the kernel
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 64) { tab_hi[tid] = (f2){(float)(tid * 256), 1.0f}; tab_lo[tid] = (f2){1.0f, (float)(tid * 256)}; }
    for (int i = tid; i < 36 * 1024; i += 512) big[i] = (float)i;
    __syncthreads();
    unsigned nA = 0, nB = 0, nC = 0;
    f2 g = {1.0f + 0.001f * tid, 2.0f + 0.003f * tid};
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const f2 p = tab_hi[(it + (tid >> 6)) & 63];
        const f2 q = tab_lo[(it + (tid >> 6)) & 63];
        // some transcendental / packed traffic in front, like the gradient phase of the kernel
        const float e = __builtin_amdgcn_exp2f(-g.x * 0.01f), r = __builtin_amdgcn_rcpf(1.0f + e);
        acc += r + big[(it * 64 + lane) % (36 * 1024)];
        f2 dA, dB, dC;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(dA) : "v"(p), "v"(g));
        // (one asm block, fixed temporaries: the copy and the multiply are adjacent, as in the kernel)
        asm volatile("v_mov_b32 v200, %2\n\tv_pk_mul_f32 %0, v[200:201], %1 op_sel_hi:[0,1]" : "=v"(dB) : "v"(g), "v"(p.y) : "v200", "v201");
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(dC) : "v"(q), "v"(g));
        nA += (dA.x != g.x) | (dA.y != g.y);
        nB += (dB.x != g.x) | (dB.y != g.y);
        nC += (dC.x != g.x) | (dC.y != g.y);
        g.x += 0.5f; g.y += 0.25f;
        if (g.x > 1000.f) { g.x -= 999.f; g.y -= 499.f; }
    }
    if (acc == 12345.678f) nA += 1000000;                              // keep `acc` alive
    atomicAdd(&bad[0 * 4 + (lane >> 4)], nA);
    atomicAdd(&bad[1 * 4 + (lane >> 4)], nB);
    atomicAdd(&bad[2 * 4 + (lane >> 4)], nC);
    if (tid == 0) atomicAdd(&bad[12], 1u);                             // workgroups that ran
}

int main() {
    unsigned* d; unsigned h[13] = {0};
    hipMalloc(&d, sizeof(h));
    hipMemset(d, 0, sizeof(h));
    const int iters = 20000, wgs = 2048;
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(512), 0, 0, d, iters);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s; %u of %d workgroups ran x 512 lanes x %d products per form; mismatches by 16-lane group [0-15, 16-31, 32-47, 48-63]\n", hipGetErrorString(e), h[12], wgs, iters);
    const char* name[3] = {"A op_sel:[1,0]          ", "B v_mov + op_sel_hi:[0,1]", "C op_sel_hi:[0,1]        "};
    for (int f = 0; f < 3; ++f) printf("  %s  %u %u %u %u\n", name[f], h[4 * f], h[4 * f + 1], h[4 * f + 2], h[4 * f + 3]);
    return 0;
}
