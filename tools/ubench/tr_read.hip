// ds_read_b64_tr_b16 semantics probe (gfx950): every lane passes the address of "its" 4 consecutive 16-bit elements
// (lane l -> lds[4 l .. 4 l + 3]); prints what each lane receives.  hipcc --offload-arch=gfx950 tr_read.hip -o tr_read && ./tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %3d %3d %3d %3d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l % 4 == 3) ? "\n" : "   ");
    return 0;
}
