// K13  dm_repack -- the operator boundary's layout change, on the device.
//
// The reference hands its operators CHANNEL-MAJOR tensors: `xz` is (B, 2 Din, L) with L contiguous (block/mamba.py:333-337
// builds it as a permuted view of the (2 Din, B L) in_proj product; CrossScan keeps that layout, block/mamba.py:31-45), and
// mamba_inner_fn / selective_scan_fn / causal_conv1d_fn take and return (B, D, L) (block/mamba.py:346-348).  Every kernel of
// this library is token-major [b][l][d] (include/diffma_hip.h, "Conventions").  The Python mirror used to bridge the two with
// `t.transpose(1, 2).contiguous()` -- a strided ATen copy at ~1 TB/s, and its autograd mirror on the way back; this kernel is
// that repack as one HBM-bound pass in either direction:
//     to_token_major = 1:  tm[b][l][d] = cm[b][d][l]          (operands entering the operator)
//     to_token_major = 0:  cm[b][d][l] = tm[b][l][d]          (gradients / outputs leaving it in the reference's layout)
// A workgroup moves a tile of CT channels (CT elements = one 128-byte line of the token-major side) x up to 256 positions through
// LDS: the channel-major side is read / written as runs along L (16-, 8-byte or element accesses, whichever the strides allow:
// L = 196 rows of 2-byte elements are 8-byte aligned), the token-major side as whole 128-byte row pieces, 16 bytes per lane.
// Bytes: one read + one write of the tensor; no arithmetic (elements are moved as 2- or 4-byte words, any dtype).
#include "dm_common.h"

namespace dm {

constexpr int RP_LT = 256;          // positions per tile
constexpr int RP_PITCH = 258;       // LDS row pitch in elements: == 2 (mod 32), so the 8 channel groups of a wave's transposed
                                    // accesses fall on 8 different bank groups (4 * pitch == 8 mod 64 dwords)

template <typename E, int V> struct rp_vec { E e[V]; };

// V consecutive elements between global memory and registers (V * sizeof(E) is 16, 8, 4 or 2 bytes; the host vouches for alignment)
template <typename E, int V>
__device__ __forceinline__ void rp_gld(E (&r)[V], const E* src) {
    if constexpr (V * sizeof(E) == 16) *reinterpret_cast<f32x4*>(r) = *reinterpret_cast<const f32x4*>(src);
    else if constexpr (V * sizeof(E) == 8) *reinterpret_cast<f32x2*>(r) = *reinterpret_cast<const f32x2*>(src);
    else if constexpr (V * sizeof(E) == 4) *reinterpret_cast<uint32_t*>(r) = *reinterpret_cast<const uint32_t*>(src);
    else {
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = src[j];
    }
}
template <typename E, int V>
__device__ __forceinline__ void rp_gst(E* dst, const E (&r)[V]) {
    if constexpr (V * sizeof(E) == 16) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(r);
    else if constexpr (V * sizeof(E) == 8) *reinterpret_cast<f32x2*>(dst) = *reinterpret_cast<const f32x2*>(r);
    else if constexpr (V * sizeof(E) == 4) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(r);
    else {
#pragma unroll
        for (int j = 0; j < V; ++j) dst[j] = r[j];
    }
}
// V consecutive elements of one LDS row (rows are 4-byte aligned: the pitch is even): dword accesses when the run allows
template <typename E, int V>
__device__ __forceinline__ void rp_lds_st(E* row, const E (&r)[V]) {
    if constexpr (V * sizeof(E) >= 4) {
#pragma unroll
        for (int j = 0; j < (int)(V * sizeof(E) / 4); ++j) reinterpret_cast<uint32_t*>(row)[j] = reinterpret_cast<const uint32_t*>(r)[j];
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) row[j] = r[j];
    }
}
template <typename E, int V>
__device__ __forceinline__ void rp_lds_ld(E (&r)[V], const E* row) {
    if constexpr (V * sizeof(E) >= 4) {
#pragma unroll
        for (int j = 0; j < (int)(V * sizeof(E) / 4); ++j) reinterpret_cast<uint32_t*>(r)[j] = reinterpret_cast<const uint32_t*>(row)[j];
    } else {
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = row[j];
    }
}

// E: the element as a word (uint16_t / uint32_t); VL: elements per access along L (channel-major side); VC: elements per access along
// the channels (token-major side; 16 bytes, or 1 when the strides do not allow it); TM: direction (true = channel-major -> token-major)
template <typename E, int VL, int VC, bool TM>
__global__ __launch_bounds__(256) void repack_kernel(const dm_repack_args p) {
    constexpr int CT = 128 / (int)sizeof(E);          // channels per tile = one 128-byte line of a token-major row
    constexpr int NG = CT / VC;                       // channel groups per row piece
    __shared__ __attribute__((aligned(16))) E lds[CT * RP_PITCH];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * CT;
    const int l0 = blockIdx.y * RP_LT;
    const int nl = min(RP_LT, p.seqlen - l0);
    const int nc = min(CT, p.dim - c0);
    const int nv = (nl + VL - 1) / VL;                // runs per channel row (VL divides seqlen on a vector path)
    const E* src = reinterpret_cast<const E*>(p.src);
    E* dst = reinterpret_cast<E*>(p.dst);
    const int64_t cm0 = (int64_t)b * p.cm_sb + (int64_t)c0 * p.cm_sd + l0;
    const int64_t tm0 = (int64_t)b * p.tm_sb + (int64_t)l0 * p.tm_sl + c0;

    if constexpr (TM) {
        for (int i = tid; i < nc * nv; i += 256) {
            const int c = i / nv, v = i - c * nv;
            E r[VL];
            rp_gld<E, VL>(r, src + cm0 + (int64_t)c * p.cm_sd + v * VL);
            rp_lds_st<E, VL>(&lds[c * RP_PITCH + v * VL], r);
        }
        __syncthreads();
        for (int i = tid; i < nl * NG; i += 256) {
            const int l = i / NG, g = i % NG;
            if (g * VC >= nc) continue;               // dim % VC == 0 on the vector path: a group is whole or absent
            E r[VC];
#pragma unroll
            for (int j = 0; j < VC; ++j) r[j] = lds[(g * VC + j) * RP_PITCH + l];
            rp_gst<E, VC>(dst + tm0 + (int64_t)l * p.tm_sl + g * VC, r);
        }
    } else {
        for (int i = tid; i < nl * NG; i += 256) {
            const int l = i / NG, g = i % NG;
            if (g * VC >= nc) continue;
            E r[VC];
            rp_gld<E, VC>(r, src + tm0 + (int64_t)l * p.tm_sl + g * VC);
#pragma unroll
            for (int j = 0; j < VC; ++j) lds[(g * VC + j) * RP_PITCH + l] = r[j];
        }
        __syncthreads();
        for (int i = tid; i < nc * nv; i += 256) {
            const int c = i / nv, v = i - c * nv;
            E r[VL];
            rp_lds_ld<E, VL>(r, &lds[c * RP_PITCH + v * VL]);
            rp_gst<E, VL>(dst + cm0 + (int64_t)c * p.cm_sd + v * VL, r);
        }
    }
}

template <typename E, int VL, int VC>
static void launch_repack_dir(const dm_repack_args& a, dim3 grid, hipStream_t st) {
    if (a.to_token_major) hipLaunchKernelGGL((repack_kernel<E, VL, VC, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((repack_kernel<E, VL, VC, false>), grid, dim3(256), 0, st, a);
}

template <typename E>
static int launch_repack(const dm_repack_args& a, hipStream_t st) {
    constexpr int ES = (int)sizeof(E), CT = 128 / ES, V16 = 16 / ES, V8 = 8 / ES;
    const void* cm = a.to_token_major ? a.src : (const void*)a.dst;
    const void* tm = a.to_token_major ? (const void*)a.dst : a.src;
    auto cm_ok = [&](int v) { return (uintptr_t)cm % (v * ES) == 0 && a.cm_sb % v == 0 && a.cm_sd % v == 0 && a.seqlen % v == 0; };
    const bool c16 = (uintptr_t)tm % 16 == 0 && a.tm_sb % V16 == 0 && a.tm_sl % V16 == 0 && a.dim % V16 == 0;
    const int vl = cm_ok(V16) ? V16 : cm_ok(V8) ? V8 : 1;
    dim3 grid((a.dim + CT - 1) / CT, (a.seqlen + RP_LT - 1) / RP_LT, a.batch);
    if (c16) {
        if (vl == V16) launch_repack_dir<E, V16, V16>(a, grid, st);
        else if (vl == V8) launch_repack_dir<E, V8, V16>(a, grid, st);
        else launch_repack_dir<E, 1, V16>(a, grid, st);
    } else {
        if (vl == V16) launch_repack_dir<E, V16, 1>(a, grid, st);
        else if (vl == V8) launch_repack_dir<E, V8, 1>(a, grid, st);
        else launch_repack_dir<E, 1, 1>(a, grid, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_repack: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_repack(const dm_repack_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_repack: null args"); return DM_ERR_ARG; }
    const dm_repack_args& a = *args;
    if (!a.src || !a.dst) { set_error("dm_repack: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.dim <= 0 || a.seqlen <= 0) { set_error("dm_repack: non-positive size"); return DM_ERR_ARG; }
    if (a.batch > 65535 || (a.seqlen + RP_LT - 1) / RP_LT > 65535) { set_error("dm_repack: batch > 65535 or seqlen > 16.7 M"); return DM_ERR_ARG; }
    if (a.to_token_major != 0 && a.to_token_major != 1) { set_error("dm_repack: to_token_major must be 0 or 1"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return launch_repack<uint32_t>(a, st);
        case DM_BF16:
        case DM_F16: return launch_repack<uint16_t>(a, st);
        default: set_error("dm_repack: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}
