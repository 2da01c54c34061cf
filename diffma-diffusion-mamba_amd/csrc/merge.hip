// K5  dm_token_merge -- out[b][t][:] = sum_k in[k][b][idx_k[t]][:]
//
// Replaces CrossMerge / merge_permutation (block/mamba.py:29-30, 59-69) applied BEFORE out_proj, and
// is the 3-slab gradient sum of the backward pass (CrossScan.backward, block/mamba.py:47-57).
// Pure HBM streaming: (nin reads + 1 write) * s bytes per output element, 16-B vector accesses,
// whole rows are gathered so the permutation is free.
#include "dm_common.h"

namespace dm {

#ifndef DM_MERGE_NT
#define DM_MERGE_NT 0          // 1: the streamed 16-byte accesses of merge / gate_bwd carry the nt hint (developer A/B)
#endif
// 16-byte (or narrower) vector move of VEC elements
template <typename T, int VEC>
__device__ __forceinline__ void vec_ld(T (&tmp)[VEC], const T* src) {
    if constexpr (VEC * sizeof(T) == 16) {
        if (DM_MERGE_NT) *(f32x4*)tmp = __builtin_nontemporal_load((const f32x4*)src);
        else *(f32x4*)tmp = *(const f32x4*)src;
    } else if constexpr (VEC * sizeof(T) == 8) {
        *(f32x2*)tmp = *(const f32x2*)src;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) tmp[j] = src[j];
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void vec_st(T* dst, const T (&tmp)[VEC]) {
    if constexpr (VEC * sizeof(T) == 16) {
        if (DM_MERGE_NT) __builtin_nontemporal_store(*(const f32x4*)tmp, (f32x4*)dst);
        else *(f32x4*)dst = *(const f32x4*)tmp;
    } else if constexpr (VEC * sizeof(T) == 8) {
        *(f32x2*)dst = *(const f32x2*)tmp;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = tmp[j];
    }
}

// GATE: out = (sum) * silu(gate row), optional `pre` = the ungated sum (see include/diffma_hip.h)
template <typename TI, typename TO, int VEC, bool GATE = false>
__global__ __launch_bounds__(256) void merge_kernel(const mix_args<dm_merge_args> pm) {
    // grid.x covers dim/VEC vectors of one row in blocks of 256 threads; grid.y = seqlen; grid.z = (launches x) batch
    const int mix = (pm.n > 1 && (int)blockIdx.z >= pm.a[0].batch) ? 1 : 0;
    const dm_merge_args& p = pm.a[mix];
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    const int b = blockIdx.z - mix * pm.a[0].batch;
    if (v * VEC >= p.dim) return;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
    for (int k = 0; k < p.nin; ++k) {
        const int r = p.row_index ? p.row_index[(int64_t)k * p.seqlen + t] : t;
        const TI* src = (const TI*)p.in + (int64_t)k * p.in_sk + (int64_t)b * p.in_sb + (int64_t)r * p.in_sl + (int64_t)v * VEC;
        alignas(16) TI tmp[VEC];
        if ((VEC == 4 && sizeof(TI) == 4) || (VEC == 8 && sizeof(TI) == 2)) {
            vec_ld<TI, VEC>(tmp, src);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) tmp[j] = src[j];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += io<TI>::ld(&tmp[j]);
    }
    if constexpr (GATE) {
        alignas(16) TI zv[VEC];
        vec_ld<TI, VEC>(zv, (const TI*)p.gate + (int64_t)b * p.g_sb + (int64_t)t * p.g_sl + (int64_t)v * VEC);
        if (p.pre) {
            alignas(16) TI pv[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) io<TI>::st(&pv[j], acc[j]);
            vec_st<TI, VEC>((TI*)p.pre + (int64_t)b * p.p_sb + (int64_t)t * p.p_sl + (int64_t)v * VEC, pv);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = io<TI>::ld(&pv[j]);      // gate the ROUNDED sum: the backward differentiates what it reads back
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] *= silu_f(io<TI>::ld(&zv[j]));
    }
    alignas(16) TO outv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) io<TO>::st(&outv[j], acc[j]);
    TO* dst = (TO*)p.out + (int64_t)b * p.o_sb + (int64_t)t * p.o_sl + (int64_t)v * VEC;
    if (VEC * sizeof(TO) == 16) {
        vec_st<TO, VEC>(dst, outv);
    } else if (VEC * sizeof(TO) == 8) {
        *(f32x2*)dst = *(const f32x2*)outv;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = outv[j];
    }
}

template <typename TI, typename TO>
static int launch_merge(const dm_merge_args& a, hipStream_t st) {
    constexpr int VECMAX = 16 / sizeof(TI);
    bool vec_ok = (a.dim % VECMAX == 0) && (a.in_sk % VECMAX == 0) && (a.in_sb % VECMAX == 0) &&
                  (a.in_sl % VECMAX == 0) && (a.o_sb % VECMAX == 0) && (a.o_sl % VECMAX == 0) &&
                  (((uintptr_t)a.in) % 16 == 0) && (((uintptr_t)a.out) % 16 == 0);
    if (a.gate) vec_ok = vec_ok && (a.g_sb % VECMAX == 0) && (a.g_sl % VECMAX == 0) && (((uintptr_t)a.gate) % 16 == 0);
    if (a.pre) vec_ok = vec_ok && (a.p_sb % VECMAX == 0) && (a.p_sl % VECMAX == 0) && (((uintptr_t)a.pre) % 16 == 0);
    unsigned gz;
    const mix_args<dm_merge_args> m = mix_make(a, gz);
    if (vec_ok) {
        dim3 grid((a.dim / VECMAX + 255) / 256, a.seqlen, a.batch * gz);
        if (a.gate) hipLaunchKernelGGL((merge_kernel<TI, TO, VECMAX, true>), grid, dim3(256), 0, st, m);
        else hipLaunchKernelGGL((merge_kernel<TI, TO, VECMAX>), grid, dim3(256), 0, st, m);
    } else {
        dim3 grid((a.dim + 255) / 256, a.seqlen, a.batch * gz);
        if (a.gate) hipLaunchKernelGGL((merge_kernel<TI, TO, 1, true>), grid, dim3(256), 0, st, m);
        else hipLaunchKernelGGL((merge_kernel<TI, TO, 1>), grid, dim3(256), 0, st, m);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_token_merge: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_token_merge(const dm_merge_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_token_merge: null args"); return DM_ERR_ARG; }
    const dm_merge_args& a = *args;
    if (!a.in || !a.out) { set_error("dm_token_merge: null tensor pointer"); return DM_ERR_ARG; }
    if (a.nin <= 0 || a.batch <= 0 || a.seqlen <= 0 || a.dim <= 0) { set_error("dm_token_merge: non-positive size"); return DM_ERR_ARG; }
    if (a.batch * DM_MAX_MIX > 65535 || a.seqlen > 65535) { set_error("dm_token_merge: batch > 32767 or seqlen > 65535"); return DM_ERR_ARG; }
    if (a.pre && !a.gate) { set_error("dm_token_merge: pre is only written by the gated form (gate != NULL)"); return DM_ERR_ARG; }
    if (a.gate && a.io_dtype != a.out_dtype) { set_error("dm_token_merge: the gated form keeps one dtype (in, gate, pre, out)"); return DM_ERR_DTYPE; }
    hipStream_t st = (hipStream_t)stream;
    if (a.io_dtype == DM_F32 && a.out_dtype == DM_F32) return launch_merge<float, float>(a, st);
    if (a.io_dtype == DM_BF16 && a.out_dtype == DM_BF16) return launch_merge<bf16_t, bf16_t>(a, st);
    if (a.io_dtype == DM_F16 && a.out_dtype == DM_F16) return launch_merge<f16_t, f16_t>(a, st);
    if (a.io_dtype == DM_F32 && a.out_dtype == DM_BF16) return launch_merge<float, bf16_t>(a, st);
    if (a.io_dtype == DM_F32 && a.out_dtype == DM_F16) return launch_merge<float, f16_t>(a, st);
    set_error("dm_token_merge: unsupported dtype pair (%d -> %d)", a.io_dtype, a.out_dtype);
    return DM_ERR_DTYPE;
}

// ------------------------------------------------------------------------------------------------------------
// dm_gate_bwd -- backward of the hoisted gate y = pre * silu(z):  g = dy * silu(z),  dz = dy * pre * silu'(z).
// One pass over the [batch][seqlen][dim] token-order tensors (3 reads + 2 writes per element, 16-B accesses); replaces
// sigmoid / silu' / the y re-accumulation / the dz store inside the three per-direction scan backwards and the 3-slab dz merge.
// ------------------------------------------------------------------------------------------------------------
namespace dm {
template <typename T, int VEC>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const mix_args<dm_gate_bwd_args> pm) {
    const int mix = (pm.n > 1 && (int)blockIdx.z >= pm.a[0].batch) ? 1 : 0;
    const dm_gate_bwd_args& p = pm.a[mix];
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    const int b = blockIdx.z - mix * pm.a[0].batch;
    if (v * VEC >= p.dim) return;
    const int64_t c = (int64_t)v * VEC;
    alignas(16) T dyv[VEC], zv[VEC], pv[VEC], gv[VEC], dzv[VEC];
    vec_ld<T, VEC>(dyv, (const T*)p.dy + (int64_t)b * p.dy_sb + (int64_t)t * p.dy_sl + c);
    vec_ld<T, VEC>(zv, (const T*)p.z + (int64_t)b * p.z_sb + (int64_t)t * p.z_sl + c);
    vec_ld<T, VEC>(pv, (const T*)p.pre + (int64_t)b * p.p_sb + (int64_t)t * p.p_sl + c);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float dy = io<T>::ld(&dyv[j]), z = io<T>::ld(&zv[j]), pre = io<T>::ld(&pv[j]);
        const float sg = sigmoid_f(z);
        io<T>::st(&gv[j], dy * z * sg);
        io<T>::st(&dzv[j], dy * pre * sg * (1.0f + z * (1.0f - sg)));
    }
    vec_st<T, VEC>((T*)p.g + (int64_t)b * p.g_sb + (int64_t)t * p.g_sl + c, gv);
    vec_st<T, VEC>((T*)p.dz + (int64_t)b * p.dz_sb + (int64_t)t * p.dz_sl + c, dzv);
}

template <typename T>
static int launch_gate_bwd(const dm_gate_bwd_args& a, hipStream_t st) {
    constexpr int VECMAX = 16 / sizeof(T);
    auto al = [&](const void* ptr, int64_t sb, int64_t sl) { return (uintptr_t)ptr % 16 == 0 && sb % VECMAX == 0 && sl % VECMAX == 0; };
    const bool vec_ok = a.dim % VECMAX == 0 && al(a.dy, a.dy_sb, a.dy_sl) && al(a.z, a.z_sb, a.z_sl) && al(a.pre, a.p_sb, a.p_sl) &&
                        al(a.g, a.g_sb, a.g_sl) && al(a.dz, a.dz_sb, a.dz_sl);
    unsigned gz;
    const mix_args<dm_gate_bwd_args> m = mix_make(a, gz);
    if (vec_ok) {
        dim3 grid((a.dim / VECMAX + 255) / 256, a.seqlen, a.batch * gz);
        hipLaunchKernelGGL((gate_bwd_kernel<T, VECMAX>), grid, dim3(256), 0, st, m);
    } else {
        dim3 grid((a.dim + 255) / 256, a.seqlen, a.batch * gz);
        hipLaunchKernelGGL((gate_bwd_kernel<T, 1>), grid, dim3(256), 0, st, m);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gate_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
}  // namespace dm

extern "C" int dm_gate_bwd(const dm_gate_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gate_bwd: null args"); return DM_ERR_ARG; }
    const dm_gate_bwd_args& a = *args;
    if (!a.dy || !a.z || !a.pre || !a.g || !a.dz) { set_error("dm_gate_bwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.seqlen <= 0 || a.dim <= 0) { set_error("dm_gate_bwd: non-positive size"); return DM_ERR_ARG; }
    if (a.batch * DM_MAX_MIX > 65535 || a.seqlen > 65535) { set_error("dm_gate_bwd: batch > 32767 or seqlen > 65535"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return launch_gate_bwd<float>(a, st);
        case DM_BF16: return launch_gate_bwd<bf16_t>(a, st);
        case DM_F16: return launch_gate_bwd<f16_t>(a, st);
        default: set_error("dm_gate_bwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}

// ------------------------------------------------------------------------------------------------------------
// dm_colsum_f32 -- out[c] = sum_r in[r][c] for a tall fp32 matrix (the per-sequence partial rows of dA / dD / dbias:
// [nseq][dim*dstate]).  ATen's outer-dimension reduction reads this shape at < 1 TB/s (107 us for 100 MB); here
// a workgroup owns 256 columns, its 16 waves stride over the rows with 16-B loads and meet in LDS.  Deterministic.
// ------------------------------------------------------------------------------------------------------------
namespace dm {
constexpr int CS_WAVES = 16;
__global__ __launch_bounds__(64 * CS_WAVES) void colsum_kernel(const mix_args<dm_colsum_args> pm) {
    const dm_colsum_args& pa = pm.a[blockIdx.z];
    const float* __restrict__ in = pa.in;
    float* __restrict__ out = pa.out;
    const int R = (int)pa.rows, C = (int)pa.cols;
    __shared__ float lds[CS_WAVES][WAVE * 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = (blockIdx.x * WAVE + lane) * 4;
    const bool ok = c < C;                               // C % 4 == 0
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* p = in + (ok ? c : 0);
    int r = wave;
    for (; r + 3 * CS_WAVES < R; r += 4 * CS_WAVES) {     // 4 independent 16-B loads in flight per lane
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + (int64_t)r * C);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + (int64_t)(r + CS_WAVES) * C);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)(r + 2 * CS_WAVES) * C);
        const f32x4 e = *reinterpret_cast<const f32x4*>(p + (int64_t)(r + 3 * CS_WAVES) * C);
        acc += (a + b) + (d + e);
    }
    for (; r < R; r += CS_WAVES) acc += *reinterpret_cast<const f32x4*>(p + (int64_t)r * C);
    *reinterpret_cast<f32x4*>(&lds[wave][lane * 4]) = acc;
    __syncthreads();
    if (wave == 0 && ok) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < CS_WAVES; ++w) t += *reinterpret_cast<const f32x4*>(&lds[w][lane * 4]);
        *reinterpret_cast<f32x4*>(out + c) = t;
    }
}
}  // namespace dm

extern "C" int dm_colsum_f32(const dm_colsum_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_colsum_f32: null args"); return DM_ERR_ARG; }
    const float* in = args->in;
    float* out = args->out;
    const int64_t rows = args->rows, cols = args->cols;
    if (!in || !out) { set_error("dm_colsum_f32: null tensor pointer"); return DM_ERR_ARG; }
    if (rows <= 0 || cols <= 0 || cols % 4 != 0 || rows > 0x7fffffff || cols > 0x7fffffff) {
        set_error("dm_colsum_f32: rows, cols must be positive and cols a multiple of 4"); return DM_ERR_ARG;
    }
    if (((uintptr_t)in % 16) || ((uintptr_t)out % 16)) { set_error("dm_colsum_f32: tensors must be 16-byte aligned"); return DM_ERR_LAYOUT; }
    unsigned gz;
    const mix_args<dm_colsum_args> m = mix_make(*args, gz);
    dim3 grid((unsigned)((cols / 4 + WAVE - 1) / WAVE), 1, gz);
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(64 * CS_WAVES), 0, (hipStream_t)stream, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_colsum_f32: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

// ------------------------------------------------------------------------------------------------------------
// dm_sum_partials -- out[m][c] = sum_w in[m][w][c], converted and placed with a row stride (see include/diffma_hip.h).
// One thread per 4 output columns; the nw partial rows of an output row are 16-B loads nw*cols floats apart.
// ------------------------------------------------------------------------------------------------------------
namespace dm {
template <typename TO>
__global__ __launch_bounds__(256) void sum_partials_kernel(const mix_args<dm_sum_partials_args> pm) {
    const dm_sum_partials_args& p = pm.a[blockIdx.z];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;           // (row, 4-column group)
    const int cg = p.cols / 4;
    if (q >= p.rows * cg) return;
    const int64_t m = q / cg;
    const int c = (int)(q - m * cg) * 4;
    const float* src = p.in + (m * p.nw) * p.cols + c;
    f32x4 acc = *reinterpret_cast<const f32x4*>(src);
    for (int w = 1; w < p.nw; ++w) acc += *reinterpret_cast<const f32x4*>(src + (int64_t)w * p.cols);
    TO* dst = (TO*)p.out + m * p.out_sr + c;
    alignas(8) TO o[4];
    io<TO>::st(&o[0], acc.x); io<TO>::st(&o[1], acc.y); io<TO>::st(&o[2], acc.z); io<TO>::st(&o[3], acc.w);
    if constexpr (sizeof(TO) == 2) *reinterpret_cast<f32x2*>(dst) = *reinterpret_cast<const f32x2*>(o);
    else *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(o);
}
}  // namespace dm

extern "C" int dm_sum_partials(const dm_sum_partials_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_sum_partials: null args"); return DM_ERR_ARG; }
    const dm_sum_partials_args& a = *args;
    if (!a.in || !a.out) { set_error("dm_sum_partials: null tensor pointer"); return DM_ERR_ARG; }
    if (a.rows <= 0 || a.nw <= 0 || a.cols <= 0 || a.cols % 4 != 0) { set_error("dm_sum_partials: rows, nw, cols must be positive and cols a multiple of 4"); return DM_ERR_ARG; }
    const int es = a.out_dtype == DM_F32 ? 4 : 2;
    if (((uintptr_t)a.in % 16) || ((uintptr_t)a.out % (4 * es)) || (a.out_sr % 4)) { set_error("dm_sum_partials: in must be 16-byte aligned, out rows aligned to 4 elements"); return DM_ERR_LAYOUT; }
    const int64_t n = a.rows * (a.cols / 4);
    if ((n + 255) / 256 > 0x7fffffff) { set_error("dm_sum_partials: too many rows"); return DM_ERR_ARG; }
    unsigned gz;
    const mix_args<dm_sum_partials_args> m = mix_make(a, gz);
    dim3 grid((unsigned)((n + 255) / 256), 1, gz);
    hipStream_t st = (hipStream_t)stream;
    switch (a.out_dtype) {
        case DM_F32: hipLaunchKernelGGL((sum_partials_kernel<float>), grid, dim3(256), 0, st, m); break;
        case DM_BF16: hipLaunchKernelGGL((sum_partials_kernel<bf16_t>), grid, dim3(256), 0, st, m); break;
        case DM_F16: hipLaunchKernelGGL((sum_partials_kernel<f16_t>), grid, dim3(256), 0, st, m); break;
        default: set_error("dm_sum_partials: bad out_dtype %d", a.out_dtype); return DM_ERR_DTYPE;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_sum_partials: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

// ---- n congruent launches in one (the two mixers of a block at small batch; see dm_common.h mix_args) ------------------------
static inline bool same_align16(const void* x, const void* y) { return (((uintptr_t)x ^ (uintptr_t)y) & 15) == 0; }

extern "C" int dm_token_merge_n(const dm_merge_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_token_merge_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_merge_args* a) { return dm_token_merge(a, stream); },
                        [](const dm_merge_args& x, const dm_merge_args& y) {
                            return same_align16(x.in, y.in) && same_align16(x.out, y.out) && same_align16(x.gate, y.gate) &&
                                   same_align16(x.pre, y.pre) &&          // the 16-byte form is chosen per pointer
                                   mix_congruent(x, y, &dm_merge_args::in, &dm_merge_args::row_index, &dm_merge_args::out,
                                                 &dm_merge_args::gate, &dm_merge_args::pre);
                        });
}

extern "C" int dm_gate_bwd_n(const dm_gate_bwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_gate_bwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_gate_bwd_args* a) { return dm_gate_bwd(a, stream); },
                        [](const dm_gate_bwd_args& x, const dm_gate_bwd_args& y) {
                            return same_align16(x.dy, y.dy) && same_align16(x.z, y.z) && same_align16(x.pre, y.pre) &&
                                   same_align16(x.g, y.g) && same_align16(x.dz, y.dz) &&
                                   mix_congruent(x, y, &dm_gate_bwd_args::dy, &dm_gate_bwd_args::z, &dm_gate_bwd_args::pre,
                                                 &dm_gate_bwd_args::g, &dm_gate_bwd_args::dz);
                        });
}

extern "C" int dm_colsum_f32_n(const dm_colsum_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_colsum_f32_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_colsum_args* a) { return dm_colsum_f32(a, stream); },
                        [](const dm_colsum_args& x, const dm_colsum_args& y) {
                            // the single entry validates args[i] only: the second struct must have the alignment of the first
                            return same_align16(x.in, y.in) && same_align16(x.out, y.out) &&
                                   mix_congruent(x, y, &dm_colsum_args::in, &dm_colsum_args::out);
                        });
}

extern "C" int dm_sum_partials_n(const dm_sum_partials_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_sum_partials_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_sum_partials_args* a) { return dm_sum_partials(a, stream); },
                        [](const dm_sum_partials_args& x, const dm_sum_partials_args& y) {
                            return same_align16(x.in, y.in) && same_align16(x.out, y.out) &&
                                   mix_congruent(x, y, &dm_sum_partials_args::in, &dm_sum_partials_args::out);
                        });
}
