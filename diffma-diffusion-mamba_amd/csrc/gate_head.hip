// K10  dm_gate_head_fwd / dm_gate_head_bwd -- the tail of DiffMa's fusion MLP in one pass per direction.
//
// The block blends its two mixers with a per-token weight  a = attention_network(cat[x_ssm, w_ssm])  where
// attention_network = LayerNorm(2C) -> Linear(2C, C) -> SiLU -> Linear(C, 1) -> Sigmoid   (reference block/mamba_block.py:90-91,
// 111-112).  LayerNorm(cat) is dm_ln_mod_fwd and Linear(2C, C) is a GEMM; what follows is, per token row m,
//     a[m] = sigmoid( sum_c silu(h[m][c] + b1[c]) * w2[c] + b2 )                       h = the GEMM's output WITHOUT its bias
// -- as ATen ops: bias epilogue, SiLU (read + write [M, C]), a [M, C] x [C, 1] product (one more read), sigmoid; and in the
// backward silu_backward (2 reads + 1 write), a rank-1 product (1 write), a split-K product + its slab sum for dw2, and two
// column reductions for the biases (one of them a full read of [M, C]).  Here: forward = ONE read of h; backward = one read of h
// + one write of dh, with db1 / dw2 / db2 accumulated per lane on the way and left as one partial row per workgroup
// ([nblk][2C + 4] fp32, reduced by dm_colsum_f32).  A wave owns a row: 64 lanes x VEC elements x NIT.
#include "dm_common.h"

namespace dm {

constexpr int GH_WAVES = 4;

template <typename T, int VEC>
__device__ __forceinline__ void gh_ld(float (&dst)[VEC], const T* p) {
    alignas(16) T tmp[VEC];
    if constexpr (VEC * sizeof(T) == 16) *(f32x4*)tmp = *(const f32x4*)p;
    else if constexpr (VEC * sizeof(T) == 8) *(f32x2*)tmp = *(const f32x2*)p;
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) tmp[j] = p[j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) dst[j] = io<T>::ld(&tmp[j]);
}
template <typename T, int VEC>
__device__ __forceinline__ void gh_st(T* p, const float (&src)[VEC]) {
    alignas(16) T tmp[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) io<T>::st(&tmp[j], src[j]);
    if constexpr (VEC * sizeof(T) == 16) *(f32x4*)p = *(const f32x4*)tmp;
    else if constexpr (VEC * sizeof(T) == 8) *(f32x2*)p = *(const f32x2*)tmp;
    else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) p[j] = tmp[j];
    }
}

__device__ __forceinline__ float gh_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <typename T, int VEC, int NIT>
__global__ __launch_bounds__(64 * GH_WAVES) void gate_head_fwd_kernel(const dm_gate_head_args p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float b1[NIT][VEC], w2[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { b1[it][j] = 0.f; w2[it][j] = 0.f; }
        if (c < p.C) {
            if (p.b1) gh_ld<float, VEC>(b1[it], p.b1 + c);
            gh_ld<float, VEC>(w2[it], p.w2 + c);
        }
    }
    const float b2 = p.b2 ? p.b2[0] : 0.f;
    const int64_t nw = (int64_t)gridDim.x * GH_WAVES;
    for (int64_t r = (int64_t)blockIdx.x * GH_WAVES + wave; r < p.rows; r += nw) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < p.C) {
                float h[VEC];
                gh_ld<T, VEC>(h, (const T*)p.h + r * p.h_sr + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float x = h[j] + b1[it][j];
                    s += x * gh_sigmoid(x) * w2[it][j];
                }
            }
        }
        s = wave_sum_dpp(s);
        if (lane == 0) io<T>::st((T*)p.a + r, gh_sigmoid(s + b2));
    }
}

// part row of a workgroup: [db1 (C) | dw2 (C) | db2, 0, 0, 0]
template <typename T, int VEC, int NIT>
__global__ __launch_bounds__(64 * GH_WAVES) void gate_head_bwd_kernel(const dm_gate_head_args p) {
    __shared__ float lds[GH_WAVES][2 * 64 * VEC * NIT + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float b1[NIT][VEC], w2[NIT][VEC], acc1[NIT][VEC], acc2[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { b1[it][j] = 0.f; w2[it][j] = 0.f; acc1[it][j] = 0.f; acc2[it][j] = 0.f; }
        if (c < p.C) {
            if (p.b1) gh_ld<float, VEC>(b1[it], p.b1 + c);
            gh_ld<float, VEC>(w2[it], p.w2 + c);
        }
    }
    float accb = 0.f;
    const int64_t nw = (int64_t)gridDim.x * GH_WAVES;
    for (int64_t r = (int64_t)blockIdx.x * GH_WAVES + wave; r < p.rows; r += nw) {
        const float a = io<T>::ld((const T*)p.a + r);
        const float dpre = io<T>::ld((const T*)p.da + r) * a * (1.0f - a);
        accb += dpre;                                       // (identical in all lanes; lane 0's copy is the one that is stored)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < p.C) {
                float h[VEC], dh[VEC];
                gh_ld<T, VEC>(h, (const T*)p.h + r * p.h_sr + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float x = h[j] + b1[it][j];
                    const float sg = gh_sigmoid(x);
                    const float g = dpre * w2[it][j] * sg * (1.0f + x * (1.0f - sg));
                    dh[j] = g;
                    acc1[it][j] += g;
                    acc2[it][j] += dpre * x * sg;
                }
                gh_st<T, VEC>((T*)p.dh + r * p.dh_sr + c, dh);
            }
        }
    }
    constexpr int CW = 64 * VEC * NIT;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = (it * 64 + lane) * VEC + j;
            lds[wave][c] = acc1[it][j];
            lds[wave][CW + c] = acc2[it][j];
        }
    }
    if (lane == 0) lds[wave][2 * CW] = accb;
    __syncthreads();
    float* row = p.part + (int64_t)blockIdx.x * (2 * p.C + 4);
    for (int c = threadIdx.x; c < 2 * p.C + 4; c += 64 * GH_WAVES) {
        float s = 0.f;
        if (c < 2 * p.C) {
            const int src = (c < p.C) ? c : CW + (c - p.C);
#pragma unroll
            for (int w = 0; w < GH_WAVES; ++w) s += lds[w][src];
        } else if (c == 2 * p.C) {
#pragma unroll
            for (int w = 0; w < GH_WAVES; ++w) s += lds[w][2 * CW];
        }
        row[c] = s;
    }
}

template <typename T, int VEC>
static int gh_launch(const dm_gate_head_args& a, hipStream_t st, bool bwd, dim3 grid, const char* who) {
    const int need = (a.C + 64 * VEC - 1) / (64 * VEC);
#define DM_GH(NIT)                                                                                                   \
    do {                                                                                                             \
        if (bwd) hipLaunchKernelGGL((gate_head_bwd_kernel<T, VEC, NIT>), grid, dim3(64 * GH_WAVES), 0, st, a);       \
        else hipLaunchKernelGGL((gate_head_fwd_kernel<T, VEC, NIT>), grid, dim3(64 * GH_WAVES), 0, st, a);          \
    } while (0)
    if (need <= 1) DM_GH(1);
    else if (need <= 2) DM_GH(2);
    else if (need <= 4) DM_GH(4);
    else { set_error("%s: row width %d exceeds %d", who, a.C, 64 * VEC * 4); return DM_ERR_ARG; }
#undef DM_GH
    return DM_OK;
}

static int gate_head_entry(const dm_gate_head_args* args, void* stream, bool bwd) {
    const char* who = bwd ? "dm_gate_head_bwd" : "dm_gate_head_fwd";
    if (!args) { set_error("%s: null args", who); return DM_ERR_ARG; }
    const dm_gate_head_args& a = *args;
    if (!a.h || !a.w2 || !a.a) { set_error("%s: null tensor pointer", who); return DM_ERR_ARG; }
    if (bwd && (!a.da || !a.dh || !a.part)) { set_error("%s: missing backward buffer", who); return DM_ERR_ARG; }
    if (a.rows <= 0 || a.C <= 0) { set_error("%s: non-positive size", who); return DM_ERR_ARG; }
    const int es = a.io_dtype == DM_F32 ? 4 : 2;
    const int vec = 16 / es;
    if (a.C % vec != 0 || a.h_sr % vec != 0 || (bwd && a.dh_sr % vec != 0) || ((uintptr_t)a.h & 15) || (bwd && ((uintptr_t)a.dh & 15)) ||
        ((uintptr_t)a.w2 & 15) || (a.b1 && ((uintptr_t)a.b1 & 15))) {
        set_error("%s: C and the row strides must be multiples of %d elements and the tensors 16-byte aligned", who, vec);
        return DM_ERR_ARG;
    }
    if (bwd && a.nblk <= 0) { set_error("%s: nblk (rows of `part`) must be positive", who); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t need_blocks = (a.rows + GH_WAVES - 1) / GH_WAVES;
    const dim3 grid((unsigned)(bwd ? a.nblk : (need_blocks < 4096 ? need_blocks : 4096)));
    int rc;
    switch (a.io_dtype) {
        case DM_F32: rc = gh_launch<float, 4>(a, st, bwd, grid, who); break;
        case DM_BF16: rc = gh_launch<bf16_t, 8>(a, st, bwd, grid, who); break;
        case DM_F16: rc = gh_launch<f16_t, 8>(a, st, bwd, grid, who); break;
        default: set_error("%s: unsupported dtype %d", who, a.io_dtype); return DM_ERR_DTYPE;
    }
    if (rc != DM_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_gate_head_fwd(const dm_gate_head_args* args, void* stream) { return dm::gate_head_entry(args, stream, false); }
extern "C" int dm_gate_head_bwd(const dm_gate_head_args* args, void* stream) { return dm::gate_head_entry(args, stream, true); }
