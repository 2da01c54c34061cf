// K9  dm_rmsnorm_merge_fwd/bwd -- gated-RMSNorm epilogue of the Mamba-2 mixer fused with the 3-way CrossMerge:
//        out[r][:] = weight[:] * sum_k  y_k[r][:] * rsqrt(mean(y_k[r][:]^2) + eps)
// (reference block/mamba2.py:349,402-403 with norm_before_gate = False: the silu(z) gate is already applied by the scan
// kernel; RMSNorm is row-wise, so it commutes with the token permutation and the merge is a plain sum of the slabs.)
// One wave per row, the row lives in registers (C <= 4096), 16-B accesses, one HBM pass: read nslab rows + write one
// (forward), read nslab + 1 rows and write nslab rows (backward).  The weight gradient leaves the kernel as one fp32
// partial row per workgroup (RMS_ROWS_PER_BLOCK rows), summed by the caller.
#include "dm_common.h"

namespace dm {

constexpr int RMS_WAVES = 4;
constexpr int RMS_ROWS_PER_BLOCK = 16;     // backward: rows per workgroup = rows per weight-gradient partial row (4 per wave)

__device__ __forceinline__ float rms_wave_sum(float v) { return wave_sum_dpp(v); }

template <typename T>
__device__ __forceinline__ void rms_ld4(float (&dst)[4], const T* p) {
    if constexpr (sizeof(T) == 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
    } else {
        alignas(8) T tmp[4];
        *reinterpret_cast<f32x2*>(tmp) = *reinterpret_cast<const f32x2*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = io<T>::ld(&tmp[j]);
    }
}
template <typename T>
__device__ __forceinline__ void rms_st4(T* p, const float (&src)[4]) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(p) = (f32x4){src[0], src[1], src[2], src[3]};
    } else {
        alignas(8) T tmp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) io<T>::st(&tmp[j], src[j]);
        *reinterpret_cast<f32x2*>(p) = *reinterpret_cast<const f32x2*>(tmp);
    }
}

template <typename T, int NIT>
__global__ __launch_bounds__(64 * RMS_WAVES) void rmsnorm_merge_fwd_kernel(const dm_rmsnorm_merge_args p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = (int64_t)blockIdx.x * RMS_WAVES + wave;
    if (r >= p.rows) return;
    const int C = p.C;
    float acc[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[it][j] = 0.f;
    for (int k = 0; k < p.nslab; ++k) {
        const T* yr = (const T*)p.y + (int64_t)k * p.y_ss + r * p.y_sr;
        float v[NIT][4];
        float q = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < C) {
                rms_ld4<T>(v[it], yr + c);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[it][j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) q += v[it][j] * v[it][j];
        }
        const float rstd = rsqrtf(rms_wave_sum(q) / (float)C + p.eps);
        if (lane == 0) p.rstd[(int64_t)k * p.rows + r] = rstd;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[it][j] += v[it][j] * rstd;
    }
    T* orow = (T*)p.out + r * p.out_sr;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < C) {
            float w[4], o[4];
            rms_ld4<float>(w, p.weight + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc[it][j] * w[j];
            rms_st4<T>(orow + c, o);
        }
    }
}

// dy_k = rstd_k * gw - y_k * rstd_k^3 * <gw, y_k> / C,  gw = dout * weight;   dweight += dout * sum_k y_k * rstd_k
template <typename T, int NIT>
__global__ __launch_bounds__(64 * RMS_WAVES) void rmsnorm_merge_bwd_kernel(const dm_rmsnorm_merge_args p) {
    __shared__ float part_lds[RMS_WAVES][NIT * 4][WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C = p.C;
    float w[NIT][4], dwl[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < C) {
            rms_ld4<float>(w[it], p.weight + c);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[it][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) dwl[it][j] = 0.f;
    }
    const int64_t r0 = (int64_t)blockIdx.x * RMS_ROWS_PER_BLOCK;
    for (int i = wave; i < RMS_ROWS_PER_BLOCK; i += RMS_WAVES) {
        const int64_t r = r0 + i;
        if (r >= p.rows) break;
        const T* gr = (const T*)p.dout + r * p.dout_sr;
        float g[NIT][4];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < C) {
                rms_ld4<T>(g[it], gr + c);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) g[it][j] = 0.f;
            }
        }
        auto load_slab = [&](int k, float (&v)[NIT][4]) {
            const T* yr = (const T*)p.y + (int64_t)k * p.y_ss + r * p.y_sr;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 4;
                if (c < C) {
                    rms_ld4<T>(v[it], yr + c);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[it][j] = 0.f;
                }
            }
        };
        float vn[NIT][4];
        load_slab(0, vn);
        for (int k = 0; k < p.nslab; ++k) {
            T* dyr = (T*)p.dy + (int64_t)k * p.dy_ss + r * p.dy_sr;
            const float rstd = p.rstd[(int64_t)k * p.rows + r];
            float v[NIT][4];
            float dot = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[it][j] = vn[it][j];
                    dot += g[it][j] * w[it][j] * v[it][j];
                }
            if (k + 1 < p.nslab) load_slab(k + 1, vn);       // the next slab's row is in flight under this one's reduction
            const float coef = rms_wave_sum(dot) * rstd * rstd * rstd / (float)C;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = (it * 64 + lane) * 4;
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = rstd * g[it][j] * w[it][j] - v[it][j] * coef;
                    dwl[it][j] += g[it][j] * v[it][j] * rstd;
                }
                if (c < C) rms_st4<T>(dyr + c, o);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) part_lds[wave][it * 4 + j][lane] = dwl[it][j];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * 4;
            if (c < C) {
                float o[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = 0.f;
#pragma unroll
                    for (int wv = 0; wv < RMS_WAVES; ++wv) o[j] += part_lds[wv][it * 4 + j][lane];
                }
                rms_st4<float>(p.dw_part + (int64_t)blockIdx.x * C + c, o);
            }
        }
    }
}

template <typename T, int NIT>
static void rms_launch2(const dm_rmsnorm_merge_args& a, hipStream_t st, bool bwd) {
    if (bwd) {
        dim3 grid((unsigned)((a.rows + RMS_ROWS_PER_BLOCK - 1) / RMS_ROWS_PER_BLOCK));
        hipLaunchKernelGGL((rmsnorm_merge_bwd_kernel<T, NIT>), grid, dim3(64 * RMS_WAVES), 0, st, a);
    } else {
        dim3 grid((unsigned)((a.rows + RMS_WAVES - 1) / RMS_WAVES));
        hipLaunchKernelGGL((rmsnorm_merge_fwd_kernel<T, NIT>), grid, dim3(64 * RMS_WAVES), 0, st, a);
    }
}

template <typename T>
static int rms_launch(const dm_rmsnorm_merge_args& a, hipStream_t st, bool bwd) {
    const int nit = (a.C + 255) / 256;
    if (nit <= 4) rms_launch2<T, 4>(a, st, bwd);
    else if (nit <= 8) rms_launch2<T, 8>(a, st, bwd);
    else if (nit <= 12) rms_launch2<T, 12>(a, st, bwd);
    else rms_launch2<T, 16>(a, st, bwd);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_rmsnorm_merge: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

static int rms_entry(const dm_rmsnorm_merge_args* args, void* stream, bool bwd) {
    const char* who = bwd ? "dm_rmsnorm_merge_bwd" : "dm_rmsnorm_merge_fwd";
    if (!args) { set_error("%s: null args", who); return DM_ERR_ARG; }
    const dm_rmsnorm_merge_args& a = *args;
    if (!a.y || !a.weight || !a.rstd) { set_error("%s: null tensor pointer", who); return DM_ERR_ARG; }
    if (bwd ? (!a.dout || !a.dy || !a.dw_part) : !a.out) { set_error("%s: null tensor pointer", who); return DM_ERR_ARG; }
    if (a.nslab <= 0 || a.rows <= 0 || a.C <= 0) { set_error("%s: non-positive size", who); return DM_ERR_ARG; }
    if (a.C > 4096 || a.C % 4 != 0) { set_error("%s: C must be a multiple of 4 and <= 4096", who); return DM_ERR_ARG; }
    const int64_t strides[] = {a.y_ss, a.y_sr, a.out_sr, a.dout_sr, a.dy_ss, a.dy_sr};
    for (int64_t s : strides)
        if (s % 4 != 0) { set_error("%s: strides must be multiples of 4 elements", who); return DM_ERR_LAYOUT; }
    const void* ptrs[] = {a.y, a.out, a.dout, a.dy, a.weight, a.dw_part};
    for (const void* q : ptrs)
        if (q && ((uintptr_t)q % 16) != 0) { set_error("%s: tensors must be 16-byte aligned", who); return DM_ERR_LAYOUT; }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return rms_launch<float>(a, st, bwd);
        case DM_BF16: return rms_launch<bf16_t>(a, st, bwd);
        case DM_F16: return rms_launch<f16_t>(a, st, bwd);
        default: set_error("%s: bad io_dtype %d", who, a.io_dtype); return DM_ERR_DTYPE;
    }
}

}  // namespace dm

extern "C" int dm_rmsnorm_merge_rows_per_block(void) { return dm::RMS_ROWS_PER_BLOCK; }
extern "C" int dm_rmsnorm_merge_fwd(const dm_rmsnorm_merge_args* args, void* stream) { return dm::rms_entry(args, stream, false); }
extern "C" int dm_rmsnorm_merge_bwd(const dm_rmsnorm_merge_args* args, void* stream) { return dm::rms_entry(args, stream, true); }
