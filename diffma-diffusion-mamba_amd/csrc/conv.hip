// K3/K4  dm_gather_conv1d_fwd / _bwd -- token gather + causal depthwise conv1d (+bias, +SiLU).
//
// Replaces, in one HBM pass, CrossScan's `x[:, :, order]` gathers (block/mamba.py:26-45) and
// causal_conv1d_cuda.causal_conv1d_fwd/bwd (inside mamba_inner_fn, call sites block/mamba.py:346-348;
// mathematics SURVEY.md A.1 step 2).
//
// Layout: token-major.  One lane per channel, one wave per (direction, batch, 64 channels, chunk of
// CH time steps).  The permutation is a gather of whole 256-B row segments, so it costs nothing; a
// chunk issues all CH+W-1 row loads up front (deep memory-level parallelism), then slides the W-tap
// window through registers.  Algorithmic bytes: read x once per direction + write out = 2*s B/element
// per direction (halo re-reads of W-1 rows per chunk are L2 hits).
#include <initializer_list>
#include "dm_common.h"

namespace dm {

constexpr int CONV_CH = 14;   // time steps per wave chunk (196 = 14*14)
constexpr int CONV_BWD_WAVES = 7;   // backward: waves (= consecutive chunks) per workgroup, their dw/db partials are summed in LDS

// VEC channels per lane: 16-bit I/O moves 2 channels per 32-bit access (half the memory instructions per byte;
// the 16-bit version is issue-bound: 3.8 vs 5.7 TB/s-equivalent measured against fp32 I/O).
template <typename T, int VEC> struct vio;
template <typename T> struct vio<T, 1> {
    static __device__ __forceinline__ void ld(const T* p, float (&v)[1]) { v[0] = io<T>::ld(p); }
    static __device__ __forceinline__ void st(T* p, const float (&v)[1]) { io<T>::st(p, v[0]); }
};
template <> struct vio<bf16_t, 2> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[2]) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
        v[0] = __uint_as_float(w << 16);
        v[1] = __uint_as_float(w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[2]) {
        uint32_t w;
        w = dm_cvt_pk_bf16(v[0], v[1]);
        *reinterpret_cast<uint32_t*>(p) = w;
    }
};
template <> struct vio<f16_t, 2> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ void ld(const f16_t* p, float (&v)[2]) {
        const h2 w = *reinterpret_cast<const h2*>(p);
        v[0] = (float)w.x;
        v[1] = (float)w.y;
    }
    static __device__ __forceinline__ void st(f16_t* p, const float (&v)[2]) {
        h2 w;
        w.x = (_Float16)v[0];
        w.y = (_Float16)v[1];
        *reinterpret_cast<h2*>(p) = w;
    }
};

template <typename T, typename TW, int W, bool SILU, int VEC>
__global__ __launch_bounds__(64) void conv_fwd_kernel(const mix_args<dm_conv_fwd_args> pm) {
    const int zper = pm.a[0].ndir * pm.a[0].batch;           // grid.z = n launches x (ndir * batch) sequences
    const int mix = (pm.n > 1 && (int)blockIdx.z >= zper) ? 1 : 0;
    const dm_conv_fwd_args& p = pm.a[mix];
    const int lane = threadIdx.x;
    const int d0 = blockIdx.x * WAVE * VEC;
    const int c = blockIdx.y;
    const int s = blockIdx.z - mix * zper;  // dir*batch + b
    const int dir = s / p.batch;
    const int b = s - dir * p.batch;
    const int L = p.seqlen;
    const bool active = (d0 + lane * VEC) < p.dim;          // dim % VEC == 0 (dispatch)
    const int d = active ? d0 + lane * VEC : p.dim - VEC;
    const int lbeg = c * CONV_CH;

    const T* __restrict__ xp = (const T*)p.x + (int64_t)b * p.x_sb + d;
    T* __restrict__ op = (T*)p.out + (int64_t)s * p.o_ss + d;
    const int32_t* __restrict__ idx = p.row_index ? p.row_index + (int64_t)dir * L : nullptr;

    float w[W][VEC], bias[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
        for (int j = 0; j < W; ++j) w[j][v] = io<TW>::ld((const TW*)p.weight + (int64_t)(d + v) * W + j);
        bias[v] = p.bias ? io<TW>::ld((const TW*)p.bias + d + v) : 0.0f;
    }

    // rows lbeg-(W-1) .. lbeg+CH-1 ; out-of-range rows contribute zero
    float xv[CONV_CH + W - 1][VEC];
#pragma unroll
    for (int j = 0; j < CONV_CH + W - 1; ++j) {
        int l = lbeg - (W - 1) + j;
        l = l < 0 ? 0 : (l >= L ? L - 1 : l);
        const int r = idx ? idx[l] : l;
        vio<T, VEC>::ld(xp + (int64_t)r * p.x_sl, xv[j]);
    }
#pragma unroll
    for (int j = 0; j < W - 1; ++j) {
        if (lbeg - (W - 1) + j < 0) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[j][v] = 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < CONV_CH; ++j) {
        const int l = lbeg + j;
        if (l < L) {
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[v] = bias[v];
#pragma unroll
                for (int k = 0; k < W; ++k) acc[v] += w[k][v] * xv[j + k][v];
                if (SILU) acc[v] = silu_f(acc[v]);
            }
            if (active) vio<T, VEC>::st(op + (int64_t)l * p.o_sl, acc);
        }
    }
}

// Backward: g[m] = dout[m]*act'(pre[m]);  dxs[m] = sum_j w[j]*g[m+W-1-j]  written at token idx[m];
// dw[j] += g[m]*x[idx[m-(W-1)+j]];  db += g[m].   A chunk needs pre/g on [lbeg, lbeg+CH+W-1) and x
// on [lbeg-(W-1), lbeg+CH+W-1).
template <typename T, typename TW, int W, bool SILU, int VEC>
__global__ __launch_bounds__(64 * CONV_BWD_WAVES) void conv_bwd_kernel(const mix_args<dm_conv_bwd_args> pm) {
    const int zper = pm.a[0].ndir * pm.a[0].batch;
    const int mix = (pm.n > 1 && (int)blockIdx.z >= zper) ? 1 : 0;
    const dm_conv_bwd_args& p = pm.a[mix];
    __shared__ float part_lds[CONV_BWD_WAVES][(W + 1) * VEC][WAVE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform on purpose: keeps the chunk's row arithmetic scalar
    const int d0 = blockIdx.x * WAVE * VEC;
    const int c = blockIdx.y * CONV_BWD_WAVES + wave;          // chunks past the end of the sequence only contribute zeros
    const int s = blockIdx.z - mix * zper;
    const int dir = s / p.batch;
    const int b = s - dir * p.batch;
    const int L = p.seqlen;
    const bool active = (d0 + lane * VEC) < p.dim;
    const int d = active ? d0 + lane * VEC : p.dim - VEC;
    const int lbeg = c * CONV_CH;
    constexpr int NG = CONV_CH + W - 1;        // g positions lbeg .. lbeg+NG-1
    constexpr int NX = CONV_CH + 2 * (W - 1);  // x positions lbeg-(W-1) .. lbeg+NG-1

    const T* __restrict__ xp = (const T*)p.x + (int64_t)b * p.x_sb + d;
    const T* __restrict__ gp = (const T*)p.dout + (int64_t)s * p.do_ss + d;
    T* __restrict__ dxp = (T*)p.dx + (int64_t)s * p.dx_ss + d;
    const int32_t* __restrict__ idx = p.row_index ? p.row_index + (int64_t)dir * L : nullptr;

    float w[W][VEC], bias[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
        for (int j = 0; j < W; ++j) w[j][v] = io<TW>::ld((const TW*)p.weight + (int64_t)(d + v) * W + j);
        bias[v] = p.bias ? io<TW>::ld((const TW*)p.bias + d + v) : 0.0f;
    }

    float xv[NX][VEC], g[NG][VEC];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        int l = lbeg - (W - 1) + j;
        l = l < 0 ? 0 : (l >= L ? L - 1 : l);
        const int r = idx ? idx[l] : l;
        vio<T, VEC>::ld(xp + (int64_t)r * p.x_sl, xv[j]);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        int l = lbeg + j;
        l = l >= L ? L - 1 : l;
        vio<T, VEC>::ld(gp + (int64_t)l * p.do_sl, g[j]);
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int l = lbeg - (W - 1) + j;
        if (l < 0 || l >= L) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) xv[j][v] = 0.0f;
        }
    }
    float dw[W][VEC], db[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
        for (int j = 0; j < W; ++j) dw[j][v] = 0.0f;
        db[v] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int l = lbeg + j;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            float gv = 0.0f;
            if (l < L) {
                gv = g[j][v];
                if (SILU) {
                    float pre = bias[v];
#pragma unroll
                    for (int k = 0; k < W; ++k) pre += w[k][v] * xv[j + k][v];
                    const float sg = sigmoid_f(pre);
                    gv *= sg * (1.0f + pre * (1.0f - sg));
                }
                if (j < CONV_CH) {   // each position's parameter gradient is owned by exactly one chunk
#pragma unroll
                    for (int k = 0; k < W; ++k) dw[k][v] += gv * xv[j + k][v];
                    db[v] += gv;
                }
            }
            g[j][v] = gv;
        }
    }
    // ---- dx rows, scattered back to token order --------------------------------------------------------
#pragma unroll
    for (int j = 0; j < CONV_CH; ++j) {
        const int m = lbeg + j;
        if (m < L) {
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[v] = 0.0f;
#pragma unroll
                for (int k = 0; k < W; ++k) acc[v] += w[k][v] * g[j + (W - 1) - k][v];
            }
            const int r = idx ? idx[m] : m;
            if (active) vio<T, VEC>::st(dxp + (int64_t)r * p.dx_sl, acc);
        }
    }
    // parameter gradients: sum the workgroup's chunks in LDS, one partial row per (sequence, workgroup).  The barrier only
    // orders LDS (lgkmcnt): __syncthreads() would also wait for the dx stores above to be acknowledged.
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
        for (int k = 0; k < W; ++k) part_lds[wave][v * W + k][lane] = dw[k][v];
        part_lds[wave][W * VEC + v][lane] = db[v];
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0 && active) {
        float acc[(W + 1) * VEC];
#pragma unroll
        for (int i = 0; i < (W + 1) * VEC; ++i) {
            acc[i] = 0.0f;
#pragma unroll
            for (int wv = 0; wv < CONV_BWD_WAVES; ++wv) acc[i] += part_lds[wv][i][lane];
        }
        const int64_t prow = (int64_t)s * p.nchunk + blockIdx.y;
        float* dwp = p.part_ss ? p.dw_partial + prow * p.part_ss + (int64_t)d * W : p.dw_partial + (prow * p.dim + d) * W;
#pragma unroll
        for (int i = 0; i < W * VEC; ++i) dwp[i] = acc[i];
        if (p.db_partial) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) p.db_partial[(p.part_ss ? prow * p.part_ss : prow * p.dim) + d + v] = acc[W * VEC + v];
        }
    }
}

static inline bool even4(const void* ptr, std::initializer_list<int64_t> strides, int dim) {
    if (((uintptr_t)ptr & 3) || (dim & 1)) return false;
    for (int64_t st : strides) if (st & 1) return false;
    return true;
}

template <typename T, typename TW, int W, int VEC>
static void launch_conv_fwd_v(const dm_conv_fwd_args& a, hipStream_t st) {
    const int nchunk = (a.seqlen + CONV_CH - 1) / CONV_CH;
    unsigned gz;
    const mix_args<dm_conv_fwd_args> m = mix_make(a, gz);
    dim3 grid((a.dim + WAVE * VEC - 1) / (WAVE * VEC), nchunk, a.ndir * a.batch * gz), block(WAVE);
    if (a.flags & DM_FLAG_SILU)
        hipLaunchKernelGGL((conv_fwd_kernel<T, TW, W, true, VEC>), grid, block, 0, st, m);
    else
        hipLaunchKernelGGL((conv_fwd_kernel<T, TW, W, false, VEC>), grid, block, 0, st, m);
}

template <typename T, typename TW, int W>
static int launch_conv_fwd(const dm_conv_fwd_args& a, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (even4(a.x, {a.x_sb, a.x_sl}, a.dim) && even4(a.out, {a.o_ss, a.o_sl}, a.dim)) launch_conv_fwd_v<T, TW, W, 2>(a, st);
        else launch_conv_fwd_v<T, TW, W, 1>(a, st);
    } else {
        launch_conv_fwd_v<T, TW, W, 1>(a, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gather_conv1d_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T, typename TW, int W, int VEC>
static void launch_conv_bwd_v(const dm_conv_bwd_args& a, hipStream_t st) {
    unsigned gz;
    const mix_args<dm_conv_bwd_args> m = mix_make(a, gz);
    dim3 grid((a.dim + WAVE * VEC - 1) / (WAVE * VEC), a.nchunk, a.ndir * a.batch * gz), block(WAVE * CONV_BWD_WAVES);
    if (a.flags & DM_FLAG_SILU)
        hipLaunchKernelGGL((conv_bwd_kernel<T, TW, W, true, VEC>), grid, block, 0, st, m);
    else
        hipLaunchKernelGGL((conv_bwd_kernel<T, TW, W, false, VEC>), grid, block, 0, st, m);
}

template <typename T, typename TW, int W>
static int launch_conv_bwd(const dm_conv_bwd_args& a, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        if (even4(a.x, {a.x_sb, a.x_sl}, a.dim) && even4(a.dout, {a.do_ss, a.do_sl}, a.dim) && even4(a.dx, {a.dx_ss, a.dx_sl}, a.dim))
            launch_conv_bwd_v<T, TW, W, 2>(a, st);
        else
            launch_conv_bwd_v<T, TW, W, 1>(a, st);
    } else {
        launch_conv_bwd_v<T, TW, W, 1>(a, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gather_conv1d_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T, typename TW, typename Args, int (*F2)(const Args&, hipStream_t), int (*F3)(const Args&, hipStream_t),
          int (*F4)(const Args&, hipStream_t)>
static int by_width(const Args& a, hipStream_t st, const char* who) {
    switch (a.width) {
        case 2: return F2(a, st);
        case 3: return F3(a, st);
        case 4: return F4(a, st);
        default: set_error("%s: width %d not in {2,3,4}", who, a.width); return DM_ERR_ARG;
    }
}

template <typename T, typename TW>
static int conv_fwd_t(const dm_conv_fwd_args& a, hipStream_t st) {
    return by_width<T, TW, dm_conv_fwd_args, launch_conv_fwd<T, TW, 2>, launch_conv_fwd<T, TW, 3>,
                    launch_conv_fwd<T, TW, 4>>(a, st, "dm_gather_conv1d_fwd");
}
template <typename T, typename TW>
static int conv_bwd_t(const dm_conv_bwd_args& a, hipStream_t st) {
    return by_width<T, TW, dm_conv_bwd_args, launch_conv_bwd<T, TW, 2>, launch_conv_bwd<T, TW, 3>,
                    launch_conv_bwd<T, TW, 4>>(a, st, "dm_gather_conv1d_bwd");
}

}  // namespace dm

// number of dw/db partial rows per sequence the backward writes (one per workgroup of CONV_BWD_WAVES chunks)
extern "C" int dm_conv_nchunk(int seqlen) {
    const int chunks = (seqlen + dm::CONV_CH - 1) / dm::CONV_CH;
    return (chunks + dm::CONV_BWD_WAVES - 1) / dm::CONV_BWD_WAVES;
}

extern "C" int dm_gather_conv1d_fwd(const dm_conv_fwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gather_conv1d_fwd: null args"); return DM_ERR_ARG; }
    const dm_conv_fwd_args& a = *args;
    if (!a.x || !a.weight || !a.out) { set_error("dm_gather_conv1d_fwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ndir <= 0) { set_error("dm_gather_conv1d_fwd: non-positive size"); return DM_ERR_ARG; }
    if ((int64_t)a.ndir * a.batch * DM_MAX_MIX > 65535) { set_error("dm_gather_conv1d_fwd: ndir*batch > 32767"); return DM_ERR_ARG; }
    if (a.ndir > 1 && !a.row_index) { set_error("dm_gather_conv1d_fwd: ndir>1 needs row_index"); return DM_ERR_ARG; }
    if (a.x_sd != 1 || a.o_sd != 1) { set_error("dm_gather_conv1d_fwd: needs token-major tensors (channel stride 1)"); return DM_ERR_LAYOUT; }
    hipStream_t st = (hipStream_t)stream;
    const bool wf32 = a.w_dtype == DM_F32;
    if (!wf32 && a.w_dtype != a.io_dtype) { set_error("dm_gather_conv1d_fwd: w_dtype must be fp32 or io_dtype"); return DM_ERR_DTYPE; }
    switch (a.io_dtype) {
        case DM_F32: return conv_fwd_t<float, float>(a, st);
        case DM_BF16: return wf32 ? conv_fwd_t<bf16_t, float>(a, st) : conv_fwd_t<bf16_t, bf16_t>(a, st);
        case DM_F16: return wf32 ? conv_fwd_t<f16_t, float>(a, st) : conv_fwd_t<f16_t, f16_t>(a, st);
        default: set_error("dm_gather_conv1d_fwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}

extern "C" int dm_gather_conv1d_bwd(const dm_conv_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gather_conv1d_bwd: null args"); return DM_ERR_ARG; }
    const dm_conv_bwd_args& a = *args;
    if (!a.x || !a.weight || !a.dout || !a.dx || !a.dw_partial) { set_error("dm_gather_conv1d_bwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ndir <= 0) { set_error("dm_gather_conv1d_bwd: non-positive size"); return DM_ERR_ARG; }
    if ((int64_t)a.ndir * a.batch * DM_MAX_MIX > 65535) { set_error("dm_gather_conv1d_bwd: ndir*batch > 32767"); return DM_ERR_ARG; }
    if (a.ndir > 1 && !a.row_index) { set_error("dm_gather_conv1d_bwd: ndir>1 needs row_index"); return DM_ERR_ARG; }
    if (a.nchunk != dm_conv_nchunk(a.seqlen)) { set_error("dm_gather_conv1d_bwd: nchunk must be dm_conv_nchunk(seqlen)"); return DM_ERR_ARG; }
    if (a.x_sd != 1 || a.do_sd != 1 || a.dx_sd != 1) { set_error("dm_gather_conv1d_bwd: needs token-major tensors"); return DM_ERR_LAYOUT; }
    hipStream_t st = (hipStream_t)stream;
    const bool wf32 = a.w_dtype == DM_F32;
    if (!wf32 && a.w_dtype != a.io_dtype) { set_error("dm_gather_conv1d_bwd: w_dtype must be fp32 or io_dtype"); return DM_ERR_DTYPE; }
    switch (a.io_dtype) {
        case DM_F32: return conv_bwd_t<float, float>(a, st);
        case DM_BF16: return wf32 ? conv_bwd_t<bf16_t, float>(a, st) : conv_bwd_t<bf16_t, bf16_t>(a, st);
        case DM_F16: return wf32 ? conv_bwd_t<f16_t, float>(a, st) : conv_bwd_t<f16_t, f16_t>(a, st);
        default: set_error("dm_gather_conv1d_bwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}

// ---- n congruent launches in one (the two mixers of a block at small batch) ------------------------------------------------
static inline bool same_align4(const void* x, const void* y) { return (((uintptr_t)x ^ (uintptr_t)y) & 3) == 0; }

extern "C" int dm_gather_conv1d_fwd_n(const dm_conv_fwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_gather_conv1d_fwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_conv_fwd_args* a) { return dm_gather_conv1d_fwd(a, stream); },
                        [](const dm_conv_fwd_args& x, const dm_conv_fwd_args& y) {
                            return same_align4(x.x, y.x) && same_align4(x.out, y.out) &&      // the 2-channel form is chosen per pointer
                                   mix_congruent(x, y, &dm_conv_fwd_args::x, &dm_conv_fwd_args::weight, &dm_conv_fwd_args::bias,
                                                 &dm_conv_fwd_args::row_index, &dm_conv_fwd_args::out);
                        });
}

extern "C" int dm_gather_conv1d_bwd_n(const dm_conv_bwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_gather_conv1d_bwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_conv_bwd_args* a) { return dm_gather_conv1d_bwd(a, stream); },
                        [](const dm_conv_bwd_args& x, const dm_conv_bwd_args& y) {
                            return same_align4(x.x, y.x) && same_align4(x.dout, y.dout) && same_align4(x.dx, y.dx) &&
                                   mix_congruent(x, y, &dm_conv_bwd_args::x, &dm_conv_bwd_args::weight, &dm_conv_bwd_args::bias,
                                                 &dm_conv_bwd_args::row_index, &dm_conv_bwd_args::dout, &dm_conv_bwd_args::dx,
                                                 &dm_conv_bwd_args::dw_partial, &dm_conv_bwd_args::db_partial);
                        });
}
