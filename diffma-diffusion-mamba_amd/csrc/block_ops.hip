// K8  dm_ln_mod_fwd/bwd, dm_blend_fwd/bwd -- the elementwise glue of Spiral_MambaBlock.forward
// (reference block/mamba_block.py:100-115) as single HBM passes:
//   ln_mod : LayerNorm (of x or of cat[x, x2]) -> optional adaLN modulate -> optional soft-mask copy
//   blend  : x + gate * (a*xs + (1-a)*ws)
// One wave per row (a row is 512 or 1024 channels: 8 or 16 values per lane in registers, 16-B accesses), so
// the row statistics are two 6-step wave reductions and nothing is re-read.  The backward kernels give every
// block DM_LN_ROWS_PER_BLOCK consecutive rows of ONE batch so that the per-batch (dshift, dscale, dgate) and
// per-channel (dgamma, dbeta) sums leave the kernel as one fp32 partial row per block (no atomics); the
// caller adds the handful of partial rows.
#include "dm_common.h"
#include <type_traits>

namespace dm {

constexpr int LN_WAVES = 4;
constexpr int LN_MAXE = 16;     // values per lane: C <= 64 * 16

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// element c of the logical row [x | x2]
template <typename T>
__device__ __forceinline__ const T* row_src(const T* x, const T* x2, int64_t r, int64_t x_sr, int64_t x2_sr, int C1, int c) {
    return (c < C1) ? x + r * x_sr + c : x2 + r * x2_sr + (c - C1);
}

template <typename T, int VEC>
__device__ __forceinline__ void ld_vec(float (&dst)[VEC], const T* p) {
    if constexpr (VEC == 4) {
        if constexpr (sizeof(T) == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p);
            dst[0] = q.x; dst[1] = q.y; dst[2] = q.z; dst[3] = q.w;
        } else {
            alignas(8) T tmp[4];
            *reinterpret_cast<f32x2*>(tmp) = *reinterpret_cast<const f32x2*>(p);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = io<T>::ld(&tmp[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) dst[j] = io<T>::ld(p + j);
    }
}
template <typename T, int VEC>
__device__ __forceinline__ void st_vec(T* p, const float (&src)[VEC]) {
    if constexpr (VEC == 4) {
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<f32x4*>(p) = (f32x4){src[0], src[1], src[2], src[3]};
        } else {
            alignas(8) T tmp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) io<T>::st(&tmp[j], src[j]);
            *reinterpret_cast<f32x2*>(p) = *reinterpret_cast<const f32x2*>(tmp);
        }
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) io<T>::st(p + j, src[j]);
    }
}

// ------------------------------------------------------------------------------------------------------
template <typename TX, typename TY, typename TM, int VEC, int NIT>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_mod_fwd_kernel(const dm_ln_mod_args p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)p.batch * p.rows_per_batch;
    const int64_t r = (int64_t)blockIdx.x * LN_WAVES + wave;
    if (r >= rows) return;
    const int C = p.C1 + p.C2;
    const int b = (int)(r / p.rows_per_batch);
    float v[NIT][VEC];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
            ld_vec<TX, VEC>(v[it], row_src((const TX*)p.x, (const TX*)p.x2, r, p.x_sr, p.x2_sr, p.C1, c));
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[it][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += v[it][j];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float dlt = (c + j < C) ? v[it][j] - mean : 0.f;
            q += dlt * dlt;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + p.eps);
    if (lane == 0 && p.stats) { p.stats[2 * r] = mean; p.stats[2 * r + 1] = rstd; }
    const float mk = p.mask ? io<TM>::ld((const TM*)p.mask + r) : 1.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < C) {
            float g[VEC], be[VEC], sc[VEC], sh[VEC], o1[VEC], o2[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) { g[j] = 1.f; be[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f; }
            if (p.gamma) ld_vec<float, VEC>(g, p.gamma + c);
            if (p.beta) ld_vec<float, VEC>(be, p.beta + c);
            if (p.scale) {
                ld_vec<TM, VEC>(sc, (const TM*)p.scale + (int64_t)b * p.mod_sb + c);
                ld_vec<TM, VEC>(sh, (const TM*)p.shift + (int64_t)b * p.mod_sb + c);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float n = (v[it][j] - mean) * rstd * g[j] + be[j];
                o1[j] = n * (1.f + sc[j]) + sh[j];
                o2[j] = o1[j] * mk;
            }
            st_vec<TY, VEC>((TY*)p.y1 + r * p.y_sr + c, o1);
            if (p.y2) st_vec<TY, VEC>((TY*)p.y2 + r * p.y_sr + c, o2);
        }
    }
}

// TM = no_mod_t: the launch has neither shift / scale nor a mask (the LayerNorm of the fusion MLP's concatenated input): the
// modulation's registers (scale row, dshift / dscale accumulators: 48 VGPRs at C = 1024) are compiled out -- 202 -> <= 168 VGPRs, a
// third wave per SIMD for a kernel that is a chain of dependent row passes.
struct no_mod_t {};
template <typename TX, typename TY, typename TM, int VEC, int NIT>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_mod_bwd_kernel(const dm_ln_mod_args p) {
    constexpr bool MOD = !std::is_same<TM, no_mod_t>::value;
    typedef typename std::conditional<MOD, TM, float>::type TMl;      // the type the (compiled-out) modulation loads are written with
    __shared__ float lds[LN_WAVES][64 * NIT * VEC];            // one of the 4 partial-sum rows at a time (64 KB for all four
                                                                // at once capped the occupancy of this HBM-bound kernel at 2 WGs/CU)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // keep the row arithmetic scalar
    const int C = p.C1 + p.C2;
    const int rpb = p.rows_per_block > 0 ? p.rows_per_block : DM_LN_ROWS_PER_BLOCK;
    const int bpb = (p.rows_per_batch + rpb - 1) / rpb;
    const int b = blockIdx.x / bpb, blk = blockIdx.x - b * bpb;
    const int row0 = blk * rpb;
    const int row1 = min(row0 + rpb, p.rows_per_batch);
    float g[NIT][VEC], be[NIT][VEC], sc[NIT][VEC];
    float a_sh[NIT][VEC], a_sc[NIT][VEC], a_g[NIT][VEC], a_b[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { g[it][j] = 1.f; be[it][j] = 0.f; sc[it][j] = 0.f; a_sh[it][j] = a_sc[it][j] = a_g[it][j] = a_b[it][j] = 0.f; }
        if (c < C) {
            if (p.gamma) ld_vec<float, VEC>(g[it], p.gamma + c);
            if (p.beta) ld_vec<float, VEC>(be[it], p.beta + c);
            if constexpr (MOD) { if (p.scale) ld_vec<TMl, VEC>(sc[it], (const TMl*)p.scale + (int64_t)b * p.mod_sb + c); }
        }
    }
    // The gradient that is ADDED to dx (dx_add: the read-only second gradient of x; accumulate: what dx / dx2 already hold) is
    // requested at the top of the row iteration with the other operands: behind the two row reductions -- where round 2 loaded it --
    // it cost a second full memory latency per row (the wave owns one row at a time, every iteration is a dependent chain).
    // (A register prefetch of the whole next row was tried: 100 -> 166 VGPRs at C = 512, 176 -> 266 at C = 1024 -- occupancy lost.)
    const bool has_old = p.dx_add != nullptr || p.accumulate != 0;
    for (int lr = row0 + wave; lr < row1; lr += LN_WAVES) {
        const int64_t r = (int64_t)b * p.rows_per_batch + lr;
        const float mean = p.stats[2 * r], rstd = p.stats[2 * r + 1];
        float mk = 1.f;
        if constexpr (MOD) { if (p.mask) mk = io<TMl>::ld((const TMl*)p.mask + r); }
        float xh[NIT][VEC], dxh[NIT][VEC], old[NIT][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < C) {
                float xv[VEC], d1[VEC], d2[VEC];
                ld_vec<TX, VEC>(xv, row_src((const TX*)p.x, (const TX*)p.x2, r, p.x_sr, p.x2_sr, p.C1, c));
                ld_vec<TY, VEC>(d1, (const TY*)p.dy1 + r * p.y_sr + c);
                if constexpr (MOD) { if (p.dy2) ld_vec<TY, VEC>(d2, (const TY*)p.dy2 + r * p.y_sr + c); }
                if (has_old) {
                    const TX* src = p.dx_add ? (const TX*)p.dx_add + r * p.dxa_sr + c
                                             : ((c < p.C1) ? (const TX*)p.dx + r * p.dx_sr + c : (const TX*)p.dx2 + r * p.dx2_sr + (c - p.C1));
                    ld_vec<TX, VEC>(old[it], src);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    xh[it][j] = (xv[j] - mean) * rstd;
                    float gm = d1[j], gn = d1[j];
                    if constexpr (MOD) {
                        if (p.dy2) gm = d1[j] + d2[j] * mk;
                        const float n = xh[it][j] * g[it][j] + be[it][j];
                        a_sh[it][j] += gm;
                        a_sc[it][j] += gm * n;
                        gn = gm * (1.f + sc[it][j]);
                    }
                    a_g[it][j] += gn * xh[it][j];
                    a_b[it][j] += gn;
                    dxh[it][j] = gn * g[it][j];
                    s1 += dxh[it][j];
                    s2 += dxh[it][j] * xh[it][j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { xh[it][j] = 0.f; dxh[it][j] = 0.f; }
            }
        }
        const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < C) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = rstd * (dxh[it][j] - m1 - xh[it][j] * m2);
                TX* dst = (c < p.C1) ? (TX*)p.dx + r * p.dx_sr + c : (TX*)p.dx2 + r * p.dx2_sr + (c - p.C1);
                if (has_old) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] += old[it][j];
                }
                st_vec<TX, VEC>(dst, o);
            }
        }
    }
    // ---- one partial row per block and quantity: sum the 4 waves through LDS -----------------------------
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int c = (it * 64 + lane) * VEC + j;
                lds[wave][c] = (k == 0) ? (MOD ? a_sh[it][j] : 0.f) : (k == 1) ? (MOD ? a_sc[it][j] : 0.f) : (k == 2) ? a_g[it][j] : a_b[it][j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 64 * LN_WAVES) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < LN_WAVES; ++w) acc += lds[w][c];
            p.part[((int64_t)blockIdx.x * 4 + k) * C + c] = acc;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
template <typename TX, typename TS, typename TG, int VEC, int NIT>
__global__ __launch_bounds__(64 * LN_WAVES) void blend_fwd_kernel(const dm_blend_args p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rows = (int64_t)p.batch * p.rows_per_batch;
    const int64_t r = (int64_t)blockIdx.x * LN_WAVES + wave;
    if (r >= rows) return;
    const int b = (int)(r / p.rows_per_batch);
    const float a = io<TS>::ld((const TS*)p.a + r);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < p.C) {
            float xv[VEC], s1[VEC], s2[VEC], gt[VEC], o[VEC];
            ld_vec<TX, VEC>(xv, (const TX*)p.x + r * p.C + c);
            ld_vec<TS, VEC>(s1, (const TS*)p.xs + r * p.C + c);
            ld_vec<TS, VEC>(s2, (const TS*)p.ws + r * p.C + c);
            ld_vec<TG, VEC>(gt, (const TG*)p.gate + (int64_t)b * p.gate_sb + c);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = xv[j] + gt[j] * (a * s1[j] + (1.f - a) * s2[j]);
            st_vec<TX, VEC>((TX*)p.out + r * p.C + c, o);
        }
    }
}

template <typename TX, typename TS, typename TG, int VEC, int NIT>
__global__ __launch_bounds__(64 * LN_WAVES) void blend_bwd_kernel(const dm_blend_args p) {
    __shared__ float lds[LN_WAVES][64 * LN_MAXE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpb = p.rows_per_block > 0 ? p.rows_per_block : DM_LN_ROWS_PER_BLOCK;
    const int bpb = (p.rows_per_batch + rpb - 1) / rpb;
    const int b = blockIdx.x / bpb, blk = blockIdx.x - b * bpb;
    const int row0 = blk * rpb;
    const int row1 = min(row0 + rpb, p.rows_per_batch);
    float gt[NIT][VEC], acc[NIT][VEC];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { gt[it][j] = 0.f; acc[it][j] = 0.f; }
        if (c < p.C) ld_vec<TG, VEC>(gt[it], (const TG*)p.gate + (int64_t)b * p.gate_sb + c);
    }
    for (int lr = row0 + wave; lr < row1; lr += LN_WAVES) {
        const int64_t r = (int64_t)b * p.rows_per_batch + lr;
        const float a = io<TS>::ld((const TS*)p.a + r);
        float sa = 0.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < p.C) {
                float gv[VEC], s1[VEC], s2[VEC], o1[VEC], o2[VEC];
                ld_vec<TX, VEC>(gv, (const TX*)p.g + r * p.C + c);
                ld_vec<TS, VEC>(s1, (const TS*)p.xs + r * p.C + c);
                ld_vec<TS, VEC>(s2, (const TS*)p.ws + r * p.C + c);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gg = gv[j] * gt[it][j];
                    o1[j] = gg * a;
                    o2[j] = gg * (1.f - a);
                    sa += gg * (s1[j] - s2[j]);
                    acc[it][j] += gv[j] * (a * s1[j] + (1.f - a) * s2[j]);
                }
                st_vec<TS, VEC>((TS*)p.dxs + r * p.C + c, o1);
                st_vec<TS, VEC>((TS*)p.dws + r * p.C + c, o2);
            }
        }
        sa = wave_sum(sa);
        if (lane == 0) io<TS>::st((TS*)p.da + r, sa);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = (it * 64 + lane) * VEC + j;
            if (c < 64 * LN_MAXE) lds[wave][c] = acc[it][j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += 64 * LN_WAVES) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < LN_WAVES; ++w) s += lds[w][c];
        p.dgate_part[(int64_t)blockIdx.x * p.C + c] = s;
    }
}

// ---- dispatch ----------------------------------------------------------------------------------------------
template <typename F>
static int pick_shape(int C, bool vec_ok, F&& f, const char* who) {
    // f(VEC, NIT): VEC 4 with 16-B rows when every offset allows it, else scalar; NIT = ceil(C / (64*VEC)) rounded up to 1,2,4,8,16
    const int vec = vec_ok && (C % 4 == 0) ? 4 : 1;
    const int need = (C + 64 * vec - 1) / (64 * vec);
    if (need > (vec == 4 ? 4 : 16)) { set_error("%s: row width %d exceeds %d", who, C, 64 * LN_MAXE); return DM_ERR_ARG; }
    int nit = 1;
    while (nit < need) nit *= 2;
    return f(vec, nit);
}

#define DM_SHAPE_SWITCH(KERNEL, TYPES, GRID, ARGS)                                                                        \
    [&](int vec, int nit) -> int {                                                                                       \
        if (vec == 4) {                                                                                                  \
            switch (nit) {                                                                                               \
                case 1: hipLaunchKernelGGL((KERNEL<TYPES, 4, 1>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                case 2: hipLaunchKernelGGL((KERNEL<TYPES, 4, 2>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                default: hipLaunchKernelGGL((KERNEL<TYPES, 4, 4>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;         \
            }                                                                                                            \
        } else {                                                                                                         \
            switch (nit) {                                                                                               \
                case 1: hipLaunchKernelGGL((KERNEL<TYPES, 1, 1>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                case 2: hipLaunchKernelGGL((KERNEL<TYPES, 1, 2>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                case 4: hipLaunchKernelGGL((KERNEL<TYPES, 1, 4>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                case 8: hipLaunchKernelGGL((KERNEL<TYPES, 1, 8>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;          \
                default: hipLaunchKernelGGL((KERNEL<TYPES, 1, 16>), GRID, dim3(64 * LN_WAVES), 0, st, ARGS); break;        \
            }                                                                                                            \
        }                                                                                                                \
        return DM_OK;                                                                                                    \
    }
#define DM_COMMA ,

static bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

template <typename TX, typename TY, typename TM>
static int ln_launch(const dm_ln_mod_args& a, hipStream_t st, bool bwd) {
    const int C = a.C1 + a.C2;
    const int ex = (int)sizeof(TX), ey = (int)sizeof(TY);
    auto ok = [&](int64_t stride, int es) { return (stride * es) % 16 == 0; };
    bool vec_ok = al16(a.x) && ok(a.x_sr, ex) && (a.C1 % 4 == 0) && (!a.x2 || (al16(a.x2) && ok(a.x2_sr, ex))) &&
                  (!a.gamma || al16(a.gamma)) && (!a.beta || al16(a.beta)) && ok(a.y_sr, ey) &&
                  (!a.scale || (al16(a.scale) && al16(a.shift) && ok(a.mod_sb, (int)sizeof(TM))));
    if (!bwd) vec_ok = vec_ok && al16(a.y1) && (!a.y2 || al16(a.y2));
    else vec_ok = vec_ok && al16(a.dy1) && (!a.dy2 || al16(a.dy2)) && al16(a.dx) && ok(a.dx_sr, ex) && (!a.dx2 || (al16(a.dx2) && ok(a.dx2_sr, ex))) &&
                  (!a.dx_add || (al16(a.dx_add) && ok(a.dxa_sr, ex)));
    const int64_t rows = (int64_t)a.batch * a.rows_per_batch;
    if (!bwd) {
        dim3 grid((unsigned)((rows + LN_WAVES - 1) / LN_WAVES));
        return pick_shape(C, vec_ok, DM_SHAPE_SWITCH(ln_mod_fwd_kernel, TX DM_COMMA TY DM_COMMA TM, grid, a), "dm_ln_mod_fwd");
    }
    const int rpb = a.rows_per_block > 0 ? a.rows_per_block : DM_LN_ROWS_PER_BLOCK;
    const int bpb = (a.rows_per_batch + rpb - 1) / rpb;
    dim3 grid((unsigned)(a.batch * bpb));
    if (!a.scale && !a.mask && !a.dy2)
        return pick_shape(C, vec_ok, DM_SHAPE_SWITCH(ln_mod_bwd_kernel, TX DM_COMMA TY DM_COMMA no_mod_t, grid, a), "dm_ln_mod_bwd");
    return pick_shape(C, vec_ok, DM_SHAPE_SWITCH(ln_mod_bwd_kernel, TX DM_COMMA TY DM_COMMA TM, grid, a), "dm_ln_mod_bwd");
}

template <typename TX, typename TY>
static int ln_by_mod(const dm_ln_mod_args& a, hipStream_t st, bool bwd) {
    switch (a.mod_dtype) {
        case DM_F32: return ln_launch<TX, TY, float>(a, st, bwd);
        case DM_BF16: return ln_launch<TX, TY, bf16_t>(a, st, bwd);
        case DM_F16: return ln_launch<TX, TY, f16_t>(a, st, bwd);
        default: set_error("dm_ln_mod: mod_dtype must be fp32, bf16 or fp16"); return DM_ERR_DTYPE;
    }
}

static int ln_entry(const dm_ln_mod_args* args, void* stream, bool bwd) {
    const char* who = bwd ? "dm_ln_mod_bwd" : "dm_ln_mod_fwd";
    if (!args) { set_error("%s: null args", who); return DM_ERR_ARG; }
    const dm_ln_mod_args& a = *args;
    if (!a.x || (a.C2 > 0) != (a.x2 != nullptr)) { set_error("%s: x / x2 pointers inconsistent with C1, C2", who); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.rows_per_batch <= 0 || a.C1 <= 0 || a.C2 < 0) { set_error("%s: non-positive size", who); return DM_ERR_ARG; }
    if ((a.scale == nullptr) != (a.shift == nullptr)) { set_error("%s: shift and scale go together", who); return DM_ERR_ARG; }
    if (!bwd && (!a.y1 || (a.mask != nullptr) != (a.y2 != nullptr))) { set_error("%s: y1 required, y2 iff mask", who); return DM_ERR_ARG; }
    if (bwd && (!a.dy1 || !a.dx || !a.stats || !a.part || (a.C2 > 0 && !a.dx2) || (a.dy2 && !a.mask))) { set_error("%s: missing backward buffer", who); return DM_ERR_ARG; }
    if (bwd && a.dx_add && a.C2 > 0) { set_error("%s: dx_add is for the single-input form (C2 == 0)", who); return DM_ERR_ARG; }
    if (bwd && (a.rows_per_block < 0 || a.rows_per_block % LN_WAVES)) { set_error("%s: rows_per_block must be 0 or a positive multiple of %d", who, LN_WAVES); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (a.x_dtype == DM_F32 && a.y_dtype == DM_F32) rc = ln_by_mod<float, float>(a, st, bwd);
    else if (a.x_dtype == DM_F32 && a.y_dtype == DM_BF16) rc = ln_by_mod<float, bf16_t>(a, st, bwd);
    else if (a.x_dtype == DM_BF16 && a.y_dtype == DM_BF16) rc = ln_by_mod<bf16_t, bf16_t>(a, st, bwd);
    else if (a.x_dtype == DM_F32 && a.y_dtype == DM_F16) rc = ln_by_mod<float, f16_t>(a, st, bwd);      // fp16 autocast (the reference's --autocast)
    else if (a.x_dtype == DM_F16 && a.y_dtype == DM_F16) rc = ln_by_mod<f16_t, f16_t>(a, st, bwd);
    else { set_error("%s: unsupported (x_dtype, y_dtype) = (%d, %d)", who, a.x_dtype, a.y_dtype); return DM_ERR_DTYPE; }
    if (rc != DM_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename TX, typename TS, typename TG>
static int blend_launch(const dm_blend_args& a, hipStream_t st, bool bwd) {
    const int ex = (int)sizeof(TX), es = (int)sizeof(TS), eg = (int)sizeof(TG);
    const bool vec_ok = (a.C * ex) % 16 == 0 && (a.C * es) % 16 == 0 && (a.gate_sb * eg) % 16 == 0 && al16(a.xs) && al16(a.ws) && al16(a.gate) &&
                        (bwd ? (al16(a.g) && al16(a.dxs) && al16(a.dws)) : (al16(a.x) && al16(a.out)));
    const int64_t rows = (int64_t)a.batch * a.rows_per_batch;
    if (!bwd) {
        dim3 grid((unsigned)((rows + LN_WAVES - 1) / LN_WAVES));
        return pick_shape(a.C, vec_ok, DM_SHAPE_SWITCH(blend_fwd_kernel, TX DM_COMMA TS DM_COMMA TG, grid, a), "dm_blend_fwd");
    }
    const int rpb = a.rows_per_block > 0 ? a.rows_per_block : DM_LN_ROWS_PER_BLOCK;
    const int bpb = (a.rows_per_batch + rpb - 1) / rpb;
    dim3 grid((unsigned)(a.batch * bpb));
    return pick_shape(a.C, vec_ok, DM_SHAPE_SWITCH(blend_bwd_kernel, TX DM_COMMA TS DM_COMMA TG, grid, a), "dm_blend_bwd");
}

static int blend_entry(const dm_blend_args* args, void* stream, bool bwd) {
    const char* who = bwd ? "dm_blend_bwd" : "dm_blend_fwd";
    if (!args) { set_error("%s: null args", who); return DM_ERR_ARG; }
    const dm_blend_args& a = *args;
    if (!a.xs || !a.ws || !a.a || !a.gate) { set_error("%s: null tensor pointer", who); return DM_ERR_ARG; }
    if (!bwd && (!a.x || !a.out)) { set_error("%s: x/out required", who); return DM_ERR_ARG; }
    if (bwd && (!a.g || !a.dxs || !a.dws || !a.da || !a.dgate_part)) { set_error("%s: missing backward buffer", who); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.rows_per_batch <= 0 || a.C <= 0) { set_error("%s: non-positive size", who); return DM_ERR_ARG; }
    if (bwd && (a.rows_per_block < 0 || a.rows_per_block % LN_WAVES)) { set_error("%s: rows_per_block must be 0 or a positive multiple of %d", who, LN_WAVES); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    int rc;
    const int key = a.x_dtype * 100 + a.s_dtype * 10 + a.g_dtype;
    switch (key) {
        case 0: rc = blend_launch<float, float, float>(a, st, bwd); break;
        case 11: rc = blend_launch<float, bf16_t, bf16_t>(a, st, bwd); break;
        case 10: rc = blend_launch<float, bf16_t, float>(a, st, bwd); break;
        case 111: rc = blend_launch<bf16_t, bf16_t, bf16_t>(a, st, bwd); break;
        case 22: rc = blend_launch<float, f16_t, f16_t>(a, st, bwd); break;
        case 20: rc = blend_launch<float, f16_t, float>(a, st, bwd); break;
        case 222: rc = blend_launch<f16_t, f16_t, f16_t>(a, st, bwd); break;
        default: set_error("%s: unsupported dtype triple (%d,%d,%d)", who, a.x_dtype, a.s_dtype, a.g_dtype); return DM_ERR_DTYPE;
    }
    if (rc != DM_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_ln_mod_fwd(const dm_ln_mod_args* args, void* stream) { return dm::ln_entry(args, stream, false); }
extern "C" int dm_ln_mod_bwd(const dm_ln_mod_args* args, void* stream) { return dm::ln_entry(args, stream, true); }
extern "C" int dm_blend_fwd(const dm_blend_args* args, void* stream) { return dm::blend_entry(args, stream, false); }
extern "C" int dm_blend_bwd(const dm_blend_args* args, void* stream) { return dm::blend_entry(args, stream, true); }
