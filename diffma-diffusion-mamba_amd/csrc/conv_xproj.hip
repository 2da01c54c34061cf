// K3x  dm_gather_conv1d_xproj_fwd -- token gather + causal depthwise conv1d + bias + SiLU + x_proj in ONE pass.
//
// BASELINE.json's north star names this kernel ("fused causal depthwise conv1d + SiLU + x-proj").  It replaces, inside
// mamba_inner_fn (reference call sites block/mamba.py:346-348; mathematics SURVEY.md A.1 steps 2-3):
//     CrossScan gather  ->  causal_conv1d_fwd + SiLU  ->  x_dbl = x~ @ x_proj.weight^T
// The unfused path writes x~ ([ndir*B, L, D]) and the library GEMM reads all of it back for a skinny (D -> R + 2N = 64)
// product; here the 16-row x~ tile a workgroup has just produced goes to LDS as 16-bit MFMA A-fragments and the product
// rides on the otherwise idle matrix pipe of an HBM-bound kernel.
//
// Mapping (CDNA4): one 256-thread workgroup (4 waves) walks ONE gathered sequence from l = 0 to L-1 in tiles of 16 rows.
//   conv phase : thread t owns channels {512*cb + 2t, +1} (32-bit accesses); the (W-1)-row window slides through registers
//                from tile to tile, so every x row is loaded exactly once per direction (no halo re-reads);
//   x_proj     : wave w owns output columns 16w .. 16w+15; its x_proj.weight rows live in registers as MFMA B-fragments for
//                the whole sequence (KSTEPS * 4 VGPRs), the A-fragments are ds_read_b128 from the padded LDS tile,
//                v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulation, one 16 x 16 output tile per wave and row tile.
// Two LDS tile buffers alternate, so there is one barrier per tile.  Algorithmic bytes: 2*s per element and direction (read
// x, write x~) + the x_dbl rows.
#include <cstdlib>
#include <type_traits>
#include "dm_common.h"

namespace dm {

constexpr int XP_TM = 16;                 // rows per tile (MFMA M)
constexpr int XP_THREADS = 512;           // 8 waves: wave w owns output columns 16*(w&3).. and the K half (w>>2)
constexpr int XP_PAD = 8;                 // tile row padding in elements (16 B): spreads the 16 rows of an A-fragment read over the banks

typedef __bf16 xp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xp_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t xp_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct xp_mfma;
template <> struct xp_mfma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const xp_u32x4& a, const xp_u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(xp_bf16x8, a), __builtin_bit_cast(xp_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        uint32_t r;
        r = dm_cvt_pk_bf16(lo, hi);                      // (through the compiler: -3 % K3x, -5 % K4x against the inline-asm form)
        return r;
    }
    static __device__ __forceinline__ void unpack(uint32_t w, float& lo, float& hi) {
        lo = __uint_as_float(w << 16);
        hi = __uint_as_float(w & 0xffff0000u);
    }
};
template <> struct xp_mfma<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 run(const xp_u32x4& a, const xp_u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(xp_f16x8, a), __builtin_bit_cast(xp_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ void unpack(uint32_t w, float& lo, float& hi) {
        const h2 v = __builtin_bit_cast(h2, w);
        lo = (float)v.x;
        hi = (float)v.y;
    }
};

// KSTEPS = dim / 32 (MFMA K steps, even).  Thread t owns channels 2t, 2t+1 (dim <= 1024).
template <typename T, typename TW, int W, bool SILU, int KSTEPS, bool IDX>
__global__ __launch_bounds__(XP_THREADS, 4) void conv_xproj_fwd_kernel(const dm_conv_xproj_fwd_args p) {
    constexpr int D = KSTEPS * 32;
    constexpr int KH = KSTEPS / 2;                                   // K steps per wave
    constexpr int ROW = D + XP_PAD;                                  // LDS row stride in elements
    static_assert(KSTEPS % 2 == 0 && D <= 2 * XP_THREADS, "dim must be a multiple of 64 and at most 1024");
    __shared__ __attribute__((aligned(16))) uint16_t tile[2][XP_TM * ROW];
    __shared__ __attribute__((aligned(16))) f32x4 red[4][WAVE];      // partial accumulators of the upper K half

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nt = wave & 3, kh = wave >> 2;
    const int g = lane >> 4, ij = lane & 15;
    // Workgroup id -> (sample, direction) with the direction FASTEST: the ndir gathered sequences of one sample read the same x rows
    // (in different orders) and now do so at about the same time, so two of the three reads are served on chip instead of from HBM.
    // (Sequence-major order ran all samples of direction 0 first: by the time direction 1 came round, x had been flushed by the
    // 655 MB of output written in between: 257-271 -> 253-258 us.)
    const int b = (int)blockIdx.x / p.ndir;
    const int dir = (int)blockIdx.x % p.ndir;
    const int s = dir * p.batch + b;
    const int L = p.seqlen;
    const cptr<int32_t> idx = IDX ? as_const(p.row_index + (int64_t)dir * L) : nullptr;     // scalar loads
    const bool act = (D == 2 * XP_THREADS) ? true : (2 * tid < D);     // compile-time true for dim 1024
    const int c = act ? 2 * tid : 0;
    // SRD addressing (dm_common.h): the lane's channel offset in one VGPR, the wave-uniform row offset in an SGPR
    const rsrc_t r_x = make_rsrc((const T*)p.x + (int64_t)b * p.x_sb);
    const rsrc_t r_o = make_rsrc((T*)p.out + (int64_t)s * p.o_ss);
    const int vo = c * (int)sizeof(T);
    const int sl_x = (int)p.x_sl * (int)sizeof(T), sl_o = (int)p.o_sl * (int)sizeof(T);
    const rsrc_t r_xd = make_rsrc((T*)p.xdbl + (int64_t)s * L * p.xd_sr);
    const int sr_xd = (int)p.xd_sr * (int)sizeof(T);

    // ---- x_proj.weight rows of this wave's 16 output columns and K half as B-fragments:
    //      lane (g, j) holds Wx[16 nt + j][32 (kh*KH + kk) + 8g .. +7] ----
    const int ncol = p.nproj;
    const int col = nt * 16 + ij;
    const bool wave_on = nt * 16 < ncol;                             // wave-uniform
    xp_u32x4 bfrag[KH];
    {   // a bounded descriptor: columns >= nproj read as zero, no branches
        const rsrc_t r_wx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wx), 0, ncol * D * (int)sizeof(T), 0x00020000);
#pragma unroll
        for (int kk = 0; kk < KH; ++kk) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_wx, (col * D + (kh * KH + kk) * 32 + g * 8) * (int)sizeof(T), 0, 0);
            bfrag[kk] = (xp_u32x4){q[0], q[1], q[2], q[3]};
        }
    }

    // ---- conv constants and the sliding window of this thread's two channels ----
    float w[W][2], bias[2], win[W - 1][2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int j = 0; j < W; ++j) w[j][v] = io<TW>::ld((const TW*)p.weight + (int64_t)(c + v) * W + j);
        bias[v] = p.bias ? io<TW>::ld((const TW*)p.bias + c + v) : 0.0f;
#pragma unroll
        for (int j = 0; j < W - 1; ++j) win[j][v] = 0.0f;             // left zero padding
    }

    // the 16 row loads of a tile are issued in two halves, each as soon as the registers of the same half of the previous tile
    // have been consumed: the x rows of tile t+1 are in flight while tile t is convolved, projected and stored
    auto load_half = [&](int l0, int h, uint32_t(&xin)[XP_TM]) {
#pragma unroll
        for (int j = h * (XP_TM / 2); j < (h + 1) * (XP_TM / 2); ++j) {
            int l = l0 + j;
            l = l < L ? l : L - 1;
            const int r = IDX ? idx[l] : l;
            xin[j] = __builtin_amdgcn_raw_buffer_load_b32(r_x, vo, r * sl_x, 0);
        }
    };
    const int ntile = (L + XP_TM - 1) / XP_TM;
    uint32_t xin[XP_TM];
    load_half(0, 0, xin);
    load_half(0, 1, xin);
    // one half (8 rows) of a tile: convolve, round, park in the LDS tile, store x~, then refill the half's registers from the next tile
    auto conv_half = [&](int t, int h) {
        const int l0 = t * XP_TM;
        uint16_t* const tl = tile[t & 1];
#pragma unroll
        for (int j = h * (XP_TM / 2); j < (h + 1) * (XP_TM / 2); ++j) {
            const bool valid = l0 + j < L;                            // wave-uniform
            float xv[2], acc[2];
            xp_mfma<T>::unpack(xin[j], xv[0], xv[1]);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                acc[v] = bias[v];                                     // same summation order as conv_fwd_kernel: bit-identical x~
#pragma unroll
                for (int k = 0; k < W - 1; ++k) acc[v] += w[k][v] * win[k][v];
                acc[v] += w[W - 1][v] * xv[v];
                if (SILU) acc[v] = silu_f(acc[v]);
#pragma unroll
                for (int k = 0; k < W - 2; ++k) win[k][v] = win[k + 1][v];
                if (W > 1) win[W - 2][v] = xv[v];
            }
            const uint32_t pk = valid ? xp_mfma<T>::pack(acc[0], acc[1]) : 0u;
            if (act) {
                *reinterpret_cast<uint32_t*>(tl + j * ROW + c) = pk;               // the A tile the matrix pipe reads (16-bit, rounded like x~)
                if (valid) __builtin_amdgcn_raw_buffer_store_b32(pk, r_o, vo, (l0 + j) * sl_o, 0);
            }
        }
        if (t + 1 < ntile) load_half(l0 + XP_TM, h, xin);
    };
    for (int t = 0; t < ntile; ++t) {
        const int l0 = t * XP_TM;
        uint16_t* const tl = tile[t & 1];
        conv_half(t, 0);
        conv_half(t, 1);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (wave_on) {
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                const xp_u32x4 a = *reinterpret_cast<const xp_u32x4*>(tl + ij * ROW + (kh * KH + kk) * 32 + g * 8);   // A[i = ij][k .. k+7]
                acc = xp_mfma<T>::run(a, bfrag[kk], acc);
            }
            if (kh == 1) red[nt][lane] = acc;
        }
        __syncthreads();
        if (wave_on && kh == 0 && col < ncol) {                        // D[4g + r][j]: rows l0 + 4g + r of column 16 nt + j
            const f32x4 o = acc + red[nt][lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = l0 + 4 * g + r;
                if (l < L) bio<T>::st(r_xd, l * sr_xd + col * (int)sizeof(T), 0, o[r]);
            }
        }
    }
}

template <typename T, typename TW, int W, int KSTEPS>
static void launch_xp(const dm_conv_xproj_fwd_args& a, hipStream_t st) {
    dim3 grid(a.ndir * a.batch), block(XP_THREADS);
    const bool silu = (a.flags & DM_FLAG_SILU) != 0;
    if (a.row_index) {
        if (silu) hipLaunchKernelGGL((conv_xproj_fwd_kernel<T, TW, W, true, KSTEPS, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_xproj_fwd_kernel<T, TW, W, false, KSTEPS, true>), grid, block, 0, st, a);
    } else {
        if (silu) hipLaunchKernelGGL((conv_xproj_fwd_kernel<T, TW, W, true, KSTEPS, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_xproj_fwd_kernel<T, TW, W, false, KSTEPS, false>), grid, block, 0, st, a);
    }
}

template <typename T, typename TW, int W>
static int xp_by_dim(const dm_conv_xproj_fwd_args& a, hipStream_t st) {
    switch (a.dim) {
        case 1024: launch_xp<T, TW, W, 32>(a, st); break;
#ifndef DM_FAST_BUILD
        case 512: launch_xp<T, TW, W, 16>(a, st); break;
        case 256: launch_xp<T, TW, W, 8>(a, st); break;
#endif
        case 128: launch_xp<T, TW, W, 4>(a, st); break;
        default: set_error("dm_gather_conv1d_xproj_fwd: dim %d not instantiated (128, 256, 512, 1024)", a.dim); return DM_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gather_conv1d_xproj_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T, typename TW>
static int xp_by_width(const dm_conv_xproj_fwd_args& a, hipStream_t st) {
    switch (a.width) {
        case 4: return xp_by_dim<T, TW, 4>(a, st);
#ifndef DM_FAST_BUILD
        case 3: return xp_by_dim<T, TW, 3>(a, st);
        case 2: return xp_by_dim<T, TW, 2>(a, st);
#endif
        default: set_error("dm_gather_conv1d_xproj_fwd: width %d not in {2,3,4}", a.width); return DM_ERR_ARG;
    }
}

}  // namespace dm

extern "C" int dm_gather_conv1d_xproj_width_supported(int width) {
#ifdef DM_FAST_BUILD
    return width == 4;
#else
    return width >= 2 && width <= 4;
#endif
}

extern "C" int dm_gather_conv1d_xproj_supported(int dim, int nproj, int io_dtype) {
    const bool d_ok = dim == 128 || dim == 256 || dim == 512 || dim == 1024;
    return (d_ok && nproj >= 1 && nproj <= 64 && (io_dtype == DM_BF16 || io_dtype == DM_F16)) ? 1 : 0;
}

extern "C" int dm_gather_conv1d_xproj_fwd(const dm_conv_xproj_fwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gather_conv1d_xproj_fwd: null args"); return DM_ERR_ARG; }
    const dm_conv_xproj_fwd_args& a = *args;
    if (!a.x || !a.weight || !a.out || !a.wx || !a.xdbl) { set_error("dm_gather_conv1d_xproj_fwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ndir <= 0) { set_error("dm_gather_conv1d_xproj_fwd: non-positive size"); return DM_ERR_ARG; }
    if (a.ndir > 1 && !a.row_index) { set_error("dm_gather_conv1d_xproj_fwd: ndir>1 needs row_index"); return DM_ERR_ARG; }
    if (!dm_gather_conv1d_xproj_supported(a.dim, a.nproj, a.io_dtype)) {
        set_error("dm_gather_conv1d_xproj_fwd: needs 16-bit I/O, dim in {128,256,512,1024}, nproj <= 64 (got dim %d nproj %d dtype %d)", a.dim, a.nproj, a.io_dtype);
        return DM_ERR_ARG;
    }
    if (a.x_sd != 1 || a.o_sd != 1) { set_error("dm_gather_conv1d_xproj_fwd: needs token-major tensors (channel stride 1)"); return DM_ERR_LAYOUT; }
    if (((uintptr_t)a.x & 3) || (a.x_sb & 1) || (a.x_sl & 1) || ((uintptr_t)a.out & 3) || (a.o_ss & 1) || (a.o_sl & 1) || ((uintptr_t)a.wx & 15)) {
        set_error("dm_gather_conv1d_xproj_fwd: x / out need 4-byte aligned rows, wx a 16-byte aligned base");
        return DM_ERR_LAYOUT;
    }
    hipStream_t st = (hipStream_t)stream;
    const bool wf32 = a.w_dtype == DM_F32;
    if (!wf32 && a.w_dtype != a.io_dtype) { set_error("dm_gather_conv1d_xproj_fwd: w_dtype must be fp32 or io_dtype"); return DM_ERR_DTYPE; }
    if (a.io_dtype == DM_BF16) return wf32 ? xp_by_width<bf16_t, float>(a, st) : xp_by_width<bf16_t, bf16_t>(a, st);
    return wf32 ? xp_by_width<f16_t, float>(a, st) : xp_by_width<f16_t, f16_t>(a, st);
}

namespace dm {

// ====================================================================================================================
// K4x  dm_gather_conv1d_xproj_bwd -- the mirror image: the gradient entering the conv is
//          dxc[s][l][:] = du[s][l][:] + dx_dbl[s*L + l][:] @ x_proj.weight          (d x~ = dL/du + W_x^T d x_dbl)
// The unfused path materialises it with an in-place addmm (reads du, writes dxc: 2 * ndir*B*L*D*s bytes) and conv_bwd reads it
// back.  Here a workgroup walks one gathered sequence in REVERSE time in tiles of 16 rows: the 16 x 64 dx_dbl tile is
// multiplied with x_proj.weight on the matrix pipe into an fp32 LDS tile [16][dim], the conv backward of the tile's rows
// adds du to it on the fly, and everything downstream (act', dw / db accumulation, dx scatter back to token order) happens
// in the same pass.  dw / db are accumulated in registers over the WHOLE sequence: one partial row per sequence.
//   wave w owns the output channels [w * dim/8, (w+1) * dim/8) of the product (dim/128 column tiles of 16);
//   thread t owns channels 2t, 2t+1 of the convolution (32-bit accesses), like K3x.
// MERGED (DM_FLAG_DX_MERGED, round 3): the workgroup owns SAMPLE b and walks its ndir gathered sequences one after the other; dx
// of every direction goes to ONE token-order buffer [batch][seqlen][dim] -- direction 0 stores, the others read-add-store the
// rows (the thread that wrote a row element is the thread that reads it back: only its own stores have to have retired, and the
// reload bypasses the CU's L1) -- so the per-direction dx slabs and the 3-slab dm_token_merge pass of the mixer's backward
// (CrossScan.backward, block/mamba.py:47-57) disappear, and dw / db leave as one partial row per sample.  Rounding: every slab
// used to be rounded to the I/O dtype before the merge summed them; here the running sum is rounded instead, the same count.
template <typename T, typename TW, int W, bool SILU, int D, bool IDX, bool MERGED = false>
__global__ __launch_bounds__(XP_THREADS, 2) void conv_xproj_bwd_kernel(const dm_conv_xproj_bwd_args p) {
    constexpr int NTW = D / 128;                                      // column tiles (16 channels) per wave
    constexpr int ROWP = D + 4;                                       // fp32 LDS row stride: rows 4g + r of a D-fragment fall on disjoint banks
    constexpr int KP = 64;                                            // projection width (nproj): two MFMA K steps
    static_assert(D % 128 == 0 && D <= 2 * XP_THREADS, "dim must be a multiple of 128 and at most 1024");
    __shared__ __attribute__((aligned(16))) float ptile[XP_TM * ROWP];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, ij = lane & 15;
    const int L = p.seqlen;
    const int b = MERGED ? (int)blockIdx.x : (int)blockIdx.x % p.batch;
    const bool act = (D == 2 * XP_THREADS) ? true : (2 * tid < D);
    const int c = act ? 2 * tid : 0;
    constexpr int ES = (int)sizeof(T);
    const rsrc_t r_x = make_rsrc((const T*)p.x + (int64_t)b * p.x_sb);
    const rsrc_t r_wt = make_rsrc(p.wxt);                             // [dim][64]
    const int vo = c * ES;
    const int sl_x = (int)p.x_sl * ES, sl_du = (int)p.du_sl * ES, sl_dx = (int)p.dx_sl * ES, sr_xd = (int)p.xd_sr * ES;

    // x_proj.weight^T rows of this wave's channels as B-fragments, resident for the whole sequence:
    // B[k][j] = Wx[k][ch] = wxt[ch][k]; lane (g, j) holds wxt[(wave*NTW + n)*16 + j][32 kk + 8g .. +7]
    xp_u32x4 bfrag[NTW][2];
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
        const int ch = (wave * NTW + n) * 16 + ij;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_wt, (ch * KP + 32 * kk + 8 * g) * ES, 0, 0);
            bfrag[n][kk] = (xp_u32x4){q[0], q[1], q[2], q[3]};
        }
    }

    float w[W][2], bias[2], dw[W][2], db[2], gwin[W - 1][2];          // gwin[k] = g[l + 1 + k] of the rows already processed (later in time)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int j = 0; j < W; ++j) {
            w[j][v] = io<TW>::ld((const TW*)p.weight + (int64_t)(c + v) * W + j);
            dw[j][v] = 0.0f;
        }
        bias[v] = p.bias ? io<TW>::ld((const TW*)p.bias + c + v) : 0.0f;
        db[v] = 0.0f;
#pragma unroll
        for (int j = 0; j < W - 1; ++j) gwin[j][v] = 0.0f;
    }

    const int ntile = (L + XP_TM - 1) / XP_TM;
    constexpr int NXR = XP_TM + W - 1;
    const int dir0 = MERGED ? 0 : (int)blockIdx.x / p.batch, dir1 = MERGED ? p.ndir : dir0 + 1;
#pragma unroll 1
    for (int dir = dir0; dir < dir1; ++dir) {
    const int s = dir * p.batch + b;
    const bool accum = MERGED && dir > 0;                              // read-add-store into the running token-order sum
    const cptr<int32_t> idx = IDX ? as_const(p.row_index + (int64_t)dir * L) : nullptr;
    const rsrc_t r_du = make_rsrc((const T*)p.du + (int64_t)s * p.du_ss);
    const rsrc_t r_dx = make_rsrc((T*)p.dx + (int64_t)(MERGED ? b : s) * p.dx_ss);
    const rsrc_t r_xd = make_rsrc((const T*)p.dxdbl + (int64_t)s * L * p.xd_sr);
    if (MERGED) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int j = 0; j < W - 1; ++j) gwin[j][v] = 0.0f;         // a new sequence: no later rows yet
        }
        if (accum) __builtin_amdgcn_s_waitcnt(0);                       // this thread's stores of the previous direction have retired
    }
    // x rows l0-(W-1) .. l0+15 of a tile (xr[0 .. W-2] = halo); rows before the sequence start are masked at use
    auto load_x = [&](int l0, uint32_t(&xr)[NXR]) {
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            const int lr = l0 - (W - 1) + j;
            const int l = lr < 0 ? 0 : (lr < L ? lr : L - 1);
            const int r = IDX ? idx[l] : l;
            xr[j] = __builtin_amdgcn_raw_buffer_load_b32(r_x, vo, r * sl_x, 0);
        }
    };
    auto load_du_half = [&](int l0, int h, uint32_t(&dur)[XP_TM]) {
#pragma unroll
        for (int j = h * (XP_TM / 2); j < (h + 1) * (XP_TM / 2); ++j) {
            int l = l0 + j;
            l = l < L ? l : L - 1;
            dur[j] = __builtin_amdgcn_raw_buffer_load_b32(r_du, vo, l * sl_du, 0);
        }
    };
    // dx_dbl rows of a tile as A-fragments: lane (g, i) holds dx_dbl[l0 + i][32 kk + 8g .. +7]; rows past L read row L-1 and are zeroed
    auto load_a = [&](int l0, xp_u32x4(&af)[2]) {
        const int lr = l0 + ij;
        const int l = lr < L ? lr : L - 1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_xd, l * sr_xd + (32 * kk + 8 * g) * ES, 0, 0);
            af[kk] = (lr < L) ? (xp_u32x4){q[0], q[1], q[2], q[3]} : (xp_u32x4){0u, 0u, 0u, 0u};
        }
    };
    // the running sum's rows of a half tile (MERGED, directions after the first), requested with the du rows half a tile ahead;
    // sc1: the reload must not be served from a line this CU cached during the previous direction's pass
    uint32_t dold[MERGED ? XP_TM : 1];
    auto load_old_half = [&](int l0, int h, uint32_t(&o)[MERGED ? XP_TM : 1]) {
        if constexpr (MERGED) {
#pragma unroll
            for (int j = h * (XP_TM / 2); j < (h + 1) * (XP_TM / 2); ++j) {
                int l = l0 + j;
                l = l < L ? l : L - 1;
                o[j] = __builtin_amdgcn_raw_buffer_load_b32(r_dx, vo, (IDX ? idx[l] : l) * sl_dx, 16);
            }
        }
    };
    uint32_t xr[NXR], xn[NXR], dur[XP_TM];
    xp_u32x4 afrag[2], anext[2];
    load_a((ntile - 1) * XP_TM, afrag);
    load_x((ntile - 1) * XP_TM, xr);
    load_du_half((ntile - 1) * XP_TM, 0, dur);
    load_du_half((ntile - 1) * XP_TM, 1, dur);
    if (MERGED && accum) {
        load_old_half((ntile - 1) * XP_TM, 0, dold);
        load_old_half((ntile - 1) * XP_TM, 1, dold);
    }

    for (int t = ntile - 1; t >= 0; --t) {
        const int l0 = t * XP_TM;
        // ---- product tile on the matrix pipe: ptile[i][n] = sum_k dx_dbl[l0 + i][k] * Wx[k][n]; no memory waits: every operand is resident ----
#pragma unroll
        for (int n = 0; n < NTW; ++n) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = xp_mfma<T>::run(afrag[0], bfrag[n][0], acc);
            acc = xp_mfma<T>::run(afrag[1], bfrag[n][1], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) ptile[(4 * g + r) * ROWP + (wave * NTW + n) * 16 + ij] = acc[r];
        }
        if (t > 0) {                                                   // the next (earlier) tile's operands: in flight during this tile's conv phase
            load_a(l0 - XP_TM, anext);
            load_x(l0 - XP_TM, xn);
        }
        __syncthreads();
        // ---- conv backward of the tile's rows, last row first ----
#pragma unroll
        for (int h = 1; h >= 0; --h) {
#pragma unroll
            for (int j = (h + 1) * (XP_TM / 2) - 1; j >= h * (XP_TM / 2); --j) {
                const int l = l0 + j;
                const bool valid = l < L;                              // wave-uniform
                const f32x2 pv = *reinterpret_cast<const f32x2*>(&ptile[j * ROWP + c]);
                float duv[2], xw[W][2];
                xp_mfma<T>::unpack(dur[j], duv[0], duv[1]);
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    xp_mfma<T>::unpack(xr[j + k], xw[k][0], xw[k][1]);
                    if (l - (W - 1) + k < 0) { xw[k][0] = 0.0f; xw[k][1] = 0.0f; }
                }
                float gv[2], dxv[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    gv[v] = valid ? (v == 0 ? pv.x : pv.y) + duv[v] : 0.0f;
                    if (SILU) {
                        float pre = bias[v];
#pragma unroll
                        for (int k = 0; k < W; ++k) pre += w[k][v] * xw[k][v];
                        const float sg = sigmoid_f(pre);
                        gv[v] *= sg * (1.0f + pre * (1.0f - sg));
                    }
#pragma unroll
                    for (int k = 0; k < W; ++k) dw[k][v] += gv[v] * xw[k][v];
                    db[v] += gv[v];
                    dxv[v] = w[W - 1][v] * gv[v];                      // dx[m] = sum_j w[j] * g[m + (W-1) - j]
#pragma unroll
                    for (int k = 0; k < W - 1; ++k) dxv[v] += w[W - 2 - k][v] * gwin[k][v];
#pragma unroll
                    for (int k = W - 2; k > 0; --k) gwin[k][v] = gwin[k - 1][v];
                    if (W > 1) gwin[0][v] = gv[v];
                }
                if (valid && act) {
                    const int r = IDX ? idx[l] : l;
                    if (MERGED && accum) {
                        float o0, o1;
                        xp_mfma<T>::unpack(dold[j], o0, o1);
                        dxv[0] += o0;
                        dxv[1] += o1;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(xp_mfma<T>::pack(dxv[0], dxv[1]), r_dx, vo, r * sl_dx, 0);
                }
            }
            if (t > 0) {
                load_du_half(l0 - XP_TM, h, dur);
                if (MERGED && accum) load_old_half(l0 - XP_TM, h, dold);
            }
        }
        __syncthreads();                                               // ptile is rewritten by the next tile's product
#pragma unroll
        for (int j = 0; j < NXR; ++j) xr[j] = xn[j];
        afrag[0] = anext[0];
        afrag[1] = anext[1];
    }
    }   // directions
    if (act) {
        const int s = MERGED ? b : (int)blockIdx.x;                   // one partial row per sample (MERGED) or per sequence
        float* dwp = p.dw_partial + (int64_t)s * (p.part_ss ? p.part_ss : (int64_t)D * W) + (int64_t)c * W;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
#pragma unroll
            for (int k = 0; k < W; ++k) dwp[v * W + k] = dw[k][v];
            if (p.db_partial) p.db_partial[(int64_t)s * (p.part_ss ? p.part_ss : (int64_t)D) + c + v] = db[v];
        }
    }
}

// ====================================================================================================================
// K4x, slab form (round 4) -- the same operator for the mixer's call pattern (DM_FLAG_DX_MERGED, width 4, SiLU, row tables) when the
// sequence is short enough for the running sum of dx to LIVE IN LDS: seqlen <= 256 (DiffMa-*/2 at 224 px: 196 tokens).
//
// The kernel above gives a workgroup a whole sample: 1024 channels x 196 tokens of dx do not fit on chip, so directions 1 and 2
// read-add-store the token-order buffer in HBM (two extra reads and two extra writes of [B, L, D]) -- 2.37 GB of HBM traffic per
// launch at the bench batch (FETCH_SIZE / WRITE_SIZE), 510-560 us.  This one cuts the sample into SLABS of 128 channels:
//   a workgroup (8 waves) works on one (sample, slab) at a time;  LDS: acc[L][128] 16-bit in TOKEN order (49 KB at L = 196), one fp32
//   product tile per wave, the slab's x_proj.weight^T as MFMA B-fragments, row tables;
//   every direction adds its dx rows into acc (the rounding of the running sum is the old kernel's, step for step), a barrier
//   separates the directions, and dx leaves ONCE, as 16-byte pieces.  Traffic: du 3x, x 1x from HBM (its two re-reads come from
//   L2: the slab is 49 KB and the workgroup is back within microseconds), dx 1x: 1.13 GB measured.
// A lane still owns one channel PAIR for the conv (64 lanes = the slab), so the 8 waves cut the gathered sequence into 8 SEGMENTS
// of ceil(L / 8) rows: the conv backward needs the gradient of the 3 rows after a segment, which the wave recomputes (28 rows of
// work for 25 at L = 196).  Rows are processed in tiles (14 or 16 rows; the MFMA M is 16) from the end of the segment to its start;
// the product tile d x_dbl . Wx goes through the wave's own LDS tile (no workgroup barrier inside a direction).  Two register sets
// of prefetched rows (x, du, d x_dbl) fit because LDS, not VGPRs, bounds the occupancy here (8 waves per CU, 256 VGPRs each): the
// loads of tile t+1 are issued before tile t's rows are touched.
// The arithmetic is written on float2 (the lane's two channels): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.
constexpr int XS_CS = 128;                 // channels per workgroup
constexpr int XS_NW = 8;                   // waves = segments of the gathered sequence
constexpr int XS_THREADS = XS_NW * WAVE;
constexpr int XS_MAXL = 256;               // acc rows (+ 1 dummy row that takes the stores of rows a wave does not own)
constexpr int XS_ROWP = XS_CS + 4;         // fp32 product tile row stride (rows 4g + r of a D-fragment on disjoint banks)
constexpr int XS_MAXDIR = 4;
constexpr int XS_SLOTS = ((XS_MAXL / XS_NW + 3 + XP_TM - 1) / XP_TM) * XP_TM;     // rows a wave walks per direction (whole tiles: 48 = 3 x 16 >= 3 x 14)
// Per (direction, row slot) of a wave.  `own` comes FIRST: it multiplies a float2 as a broadcast of the LOW dword of the pair the
// table read returns (op_sel_hi).  With acc_off first the compiler broadcasts the HIGH dword (op_sel:[1,0], or a v_mov into the low
// register right in front of the v_pk_mul), and on MI355X that form gave the last 16 lanes a stale value for their first channel
// now and then -- single rows missing from dw / db in a few channels (tools/dbg_k4x2.py; ROCm 7.2 hipcc).
#ifndef DM_K4X_REPRO
#define DM_K4X_REPRO 0       // developer: 1..5 rebuild the FAILING table layout with the multiply pinned in inline asm (tools/ubench/k4x_repro.sh)
#endif
#if DM_K4X_REPRO == 7        // control: the SHIPPED layout (own in the low dword) with the multiply pinned in inline asm as well
struct xs_row { float own; uint32_t acc_off; };
__device__ __forceinline__ f32x2 xs_mul_own_hi(const f32x2 gv, const f32x2 ent) {
    f32x2 o;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(o) : "v"(gv), "v"(ent));
    return o;
}
#elif DM_K4X_REPRO
struct xs_row { uint32_t acc_off; float own; };
// gv * own with own in the HIGH dword of the table pair: the instruction form hipcc chose for this layout, pinned, with the
// candidates for a cure around it: 1 bare, 2 s_nop 7 behind it, 3 s_waitcnt lgkmcnt(0) in front of it, 4 s_nop 7 in front of it,
// 5 the pair copied to a fresh register pair first (v_mov x2) and the multiply reading the copy
__device__ __forceinline__ f32x2 xs_mul_own_hi(const f32x2 gv, const f32x2 ent) {
    f32x2 o;
#if DM_K4X_REPRO == 1
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(o) : "v"(gv), "v"(ent));
#elif DM_K4X_REPRO == 2
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]\n\ts_nop 7" : "=v"(o) : "v"(gv), "v"(ent));
#elif DM_K4X_REPRO == 3
    asm volatile("s_waitcnt lgkmcnt(0)\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(o) : "v"(gv), "v"(ent));
#elif DM_K4X_REPRO == 4
    asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(o) : "v"(gv), "v"(ent));
#elif DM_K4X_REPRO == 8      // two plain multiplies by the high dword
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(o.x) : "v"(gv.x), "v"(ent.y));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(o.y) : "v"(gv.y), "v"(ent.y));
#elif DM_K4X_REPRO == 9      // early-clobber result: the destination pair shares no register with a source
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=&v"(o) : "v"(gv), "v"(ent));
#elif DM_K4X_REPRO == 10     // sources swapped: own's pair as src0
    asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[1,1]" : "=&v"(o) : "v"(gv), "v"(ent));
#else
    f32x2 cp;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\ts_nop 1" : "=&v"(cp.x), "=&v"(cp.y) : "v"(ent.x), "v"(ent.y));
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(o) : "v"(gv), "v"(cp));
#endif
    return o;
}
#else
struct xs_row { float own; uint32_t acc_off; };
#endif
constexpr int XS_TOKP = XS_MAXL + 32;        // tokens of rows -3 .. L-1+, padded: a tile may start 3 rows early and end past the sequence

template <typename T> struct xs_pair;      // the lane's two 16-bit channels <-> float2
template <> struct xs_pair<bf16_t> {
    static __device__ __forceinline__ f32x2 up(uint32_t w) { return (f32x2){__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
};
template <> struct xs_pair<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x2 up(uint32_t w) { return __builtin_convertvector(__builtin_bit_cast(h2, w), f32x2); }
};

__device__ __forceinline__ int64_t xs_uniform64(int64_t v) {      // a wave-uniform value back in SGPRs
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <typename T, int RT> struct xs_bufs {     // one tile's prefetched rows
    uint32_t x[RT + 3];                    // x rows l0-3 .. l0+RT-1 (gathered), the lane's channel pair
    uint32_t du[RT];
    xp_u32x4 a[2];                         // d x_dbl rows l0 .. l0+15 as MFMA A-fragments
};

// RT: rows of a tile the conv walks (the product tile always has the 16 rows of the MFMA): 16, or 14 when that wastes fewer row
// slots -- at L = 196 a wave walks 25 + 3 rows per direction = 2 x 14.
//
// PERSISTENT: the grid is one workgroup per CU (LDS allows no second one), and a workgroup keeps its SLAB while it walks through
// samples: x_proj.weight^T fragments, conv weights, the row tables are set up once, and the first tile of the next sample is
// requested during the last tile of the current one -- with one workgroup per CU nothing else would cover a workgroup's start-up
// (weights, tables, the first loads' latency) and its dx write-out.  Workgroup k: XCD k % 8, slab (k / 8) % nslab, sample stream
// (k / 8) / nslab; its samples are b = (it * nstream + stream) * 8 + xcd: the slabs of one sample run at the same time on ONE XCD, so
// the sample's d x_dbl rows (read by every slab) are fetched into one L2 and the 256-byte pieces of a du row are requested together.
// dw / db: summed over the workgroup's samples in registers, written once into the partial row of its first sample (zeros into the
// rows of the others: the caller sums the rows).
template <typename T, typename TW, int RT>
__global__ __launch_bounds__(XS_THREADS, 1) void conv_xproj_bwd_slab_kernel(const dm_conv_xproj_bwd_args p) {
    constexpr int W = 4, KP = 64, NT = XS_CS / 16, ES = (int)sizeof(T), NXR = RT + W - 1, AW = XS_CS / 2, HR = RT / 2;
    static_assert(RT % 2 == 0 && RT <= XP_TM, "whole half tiles");
    __shared__ __attribute__((aligned(16))) uint32_t accs[(XS_MAXL + 1) * AW];
    __shared__ __attribute__((aligned(16))) float ptile[XS_NW][XP_TM * XS_ROWP];
    __shared__ __attribute__((aligned(16))) xp_u32x4 bfl[NT * 2 * WAVE];          // x_proj.weight^T of the slab in MFMA B-fragment order
    __shared__ __attribute__((aligned(8))) xs_row rowtab[XS_NW][XS_MAXDIR * XS_SLOTS];
    __shared__ uint8_t xtok[XS_MAXDIR][XS_TOKP];                                  // token of gathered row l at [dir][l + 3], l = -3 .. (clamped into the sequence)

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int g = lane >> 4, ij = lane & 15;
    const int L = p.seqlen, nslab = p.dim / XS_CS;
    const int xcd = (int)blockIdx.x & 7, kq = (int)blockIdx.x >> 3;
    const int slab = kq % nslab, nstream = ((int)gridDim.x >> 3) / nslab, bstep = 8 * nstream;
    const int b_first = (kq / nslab) * 8 + xcd;
    if (b_first >= p.batch) return;                                               // (workgroup-uniform: before any barrier)
    const int c0 = slab * XS_CS, c = c0 + 2 * lane;
    {   // the running sum starts at zero: direction 0 adds like the others
        const xp_u32x4 z = {0u, 0u, 0u, 0u};
        for (int q = tid; q < L * (AW / 4); q += XS_THREADS) reinterpret_cast<xp_u32x4*>(accs)[q] = z;
    }
    for (int q = tid; q < p.ndir * XS_TOKP; q += XS_THREADS) {
        const int dir = q / XS_TOKP, lr = q % XS_TOKP - (W - 1);
        xtok[dir][q % XS_TOKP] = (uint8_t)p.row_index[dir * L + (lr < 0 ? 0 : (lr < L ? lr : L - 1))];
    }
    const rsrc_t r_wt = make_rsrc((const T*)p.wxt + (int64_t)c0 * KP);
    const int vo = 2 * lane * ES;
    const int sl_x = (int)p.x_sl * ES, sl_du = (int)p.du_sl * ES, sr_xd = (int)p.xd_sr * ES;

    // x_proj.weight^T of the slab as B-fragments, parked in LDS in fragment order (wave n fetches column tile n): lane (g, j) of
    // fragment (n, kk) holds wxt[c0 + 16 n + j][32 kk + 8g .. +7].  (In registers they are 64 VGPRs: with two sets of prefetched
    // rows the kernel then spills.)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_wt, ((wave * 16 + ij) * KP + 32 * kk + 8 * g) * ES, 0, 0);
        bfl[(wave * 2 + kk) * WAVE + lane] = (xp_u32x4){q[0], q[1], q[2], q[3]};
    }
    f32x2 w[W], bias, dw[W], db, gnext[W - 1];            // gnext[k] = gradient of row (end of the half in work) + k: the rows done before
#pragma unroll
    for (int k = 0; k < W; ++k) {
        w[k] = (f32x2){io<TW>::ld((const TW*)p.weight + (int64_t)c * W + k), io<TW>::ld((const TW*)p.weight + (int64_t)(c + 1) * W + k)};
        dw[k] = (f32x2){0.f, 0.f};
    }
    bias = p.bias ? (f32x2){io<TW>::ld((const TW*)p.bias + c), io<TW>::ld((const TW*)p.bias + c + 1)} : (f32x2){0.f, 0.f};
    db = (f32x2){0.f, 0.f};

    // this wave's segment [a, e) of every gathered sequence; every wave runs the same number of tiles (a wave whose segment lies
    // past the end works on masked rows: only short sequences have such waves)
    const int seg = (L + XS_NW - 1) / XS_NW;
    const int a = wave * seg;
    const int e = (a + seg < L) ? a + seg : L;
    const int nt = (seg + (W - 1) + RT - 1) / RT;
    float* const pt = ptile[wave];
    // where the dx of (direction, row slot) goes and whether the row counts for dw / db: this wave's rows add to their token's row of
    // acc, the rows it only recomputes (the 3 after its segment, the rest of the tile) store to a dummy row
    xs_row* const rt = rowtab[wave];
    for (int q = lane; q < p.ndir * nt * RT; q += WAVE) {
        const int dir = q / (nt * RT), l = a + q % (nt * RT);
        const bool own = l < e;
        rt[q].acc_off = (uint32_t)((own ? p.row_index[dir * L + l] : XS_MAXL) * AW * 4);
        rt[q].own = own ? 1.0f : 0.0f;
    }
    const int du_bytes = (L - 1) * sl_du + XS_CS * ES;   // a du / d x_dbl row past the end of the sequence is out of the descriptor's range:
    const int xd_bytes = L * sr_xd;                       // it loads as ZERO, and with it the gradient of the row

    // Loads of tile (sample bb, direction dir, tile t): rows l0 = a + RT t ...  tokv: the tokens of the tile's x rows, one LDS byte per
    // lane (`tokens`, read ahead of time), handed to the scalar unit lane by lane.  `live` false (after the last tile): the same
    // instructions with an out-of-range lane offset -- they load nothing, and the number of loads in flight stays what the compiler's
    // s_waitcnt bookkeeping assumes on every path (a branch around the loads makes it wait for the NEW loads wherever it has to wait
    // for an old one: no prefetch left).  Bases: 64-bit products are formed on the vector unit by this compiler; xs_uniform64 brings
    // them back to SGPRs (an address left in VGPRs costs a waterfall loop per load).
    auto tokens = [&](int dir, int t) -> int { return xtok[dir][a + RT * t + (lane < NXR ? lane : NXR - 1)]; };
    auto issue = [&](int bb, int dir, int t, bool live, int tokv, xs_bufs<T, RT>& o) {       // d x_dbl and x rows of a tile
        const int l0 = a + RT * t;
        const int vo_l = live ? vo : BIO_OOB;
        const int s = dir * p.batch + bb;
        const rsrc_t r_x = make_rsrc_2g((const T*)p.x + xs_uniform64((int64_t)bb * p.x_sb + c0));
        const rsrc_t r_xd = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)p.dxdbl + xs_uniform64((int64_t)s * L * p.xd_sr)), 0, xd_bytes, 0x00020000);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_xd, (l0 + ij) * sr_xd + (32 * kk + 8 * g) * ES + (live ? 0 : BIO_OOB), 0, 0);   // (a sum: a select on the uniform flag becomes a branch)
            o.a[kk] = (xp_u32x4){q[0], q[1], q[2], q[3]};
        }
#pragma unroll
        for (int j = NXR - 1; j >= 0; --j) o.x[j] = __builtin_amdgcn_raw_buffer_load_b32(r_x, vo_l, __builtin_amdgcn_readlane(tokv, j) * sl_x, 0);
    };
    // du is the stream that comes from HBM: its rows are requested TWO tiles ahead, half a tile at a time, into the registers the
    // tile in work has just consumed (the two register sets alternate, so the tile after next uses this one's)
    auto issue_du = [&](int bb, int dir, int t, bool live, int hh, xs_bufs<T, RT>& o) {
        const int l0 = a + RT * t;
        const int vo_l = live ? vo : BIO_OOB;
        const int s = dir * p.batch + bb;
        const rsrc_t r_du = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>((const T*)p.du + xs_uniform64((int64_t)s * p.du_ss + c0)), 0, du_bytes, 0x00020000);
#pragma unroll
        for (int j = (hh + 1) * HR - 1; j >= hh * HR; --j) o.du[j] = __builtin_amdgcn_raw_buffer_load_b32(r_du, vo_l, (l0 + j) * sl_du, 0);
    };
    // one tile: product on the matrix pipe into the wave's LDS tile, the next tile's loads, then the rows, last first
    auto step = [&](int bb, int dir, int t, bool more, int b_n, int dir_n, int t_n, bool more2, int b_2, int dir_2, int t_2,
                    xs_bufs<T, RT>& cur, xs_bufs<T, RT>& nxt) {
        const int l0 = a + RT * t;
        const xs_row* const rw = rt + (dir * nt + t) * RT;
        if (t == nt - 1) {
#pragma unroll
            for (int k = 0; k < W - 1; ++k) gnext[k] = (f32x2){0.f, 0.f};         // a new sequence: no later rows yet
        }
        const int tok_n = tokens(more ? dir_n : 0, more ? t_n : 0);
        // the next tile's d x_dbl and x rows go out at once, ahead of the product (at most 19 + 2 x 14 loads are in flight: inside the
        // 6-bit vmcnt counter, and every load is unconditional, so the compiler's waits are exact)
        issue(more ? b_n : bb, more ? dir_n : 0, more ? t_n : 0, more, tok_n, nxt);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = xp_mfma<T>::run(cur.a[0], bfl[(n * 2) * WAVE + lane], acc);
            acc = xp_mfma<T>::run(cur.a[1], bfl[(n * 2 + 1) * WAVE + lane], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[(4 * g + r) * XS_ROWP + n * 16 + ij] = acc[r];
        }
#pragma unroll
        for (int h = 1; h >= 0; --h) {
            // ---- the gradient entering the conv, rows of this half: independent of each other ----
            f32x2 gv[HR + W - 1];
#pragma unroll
            for (int k = 0; k < W - 1; ++k) gv[HR + k] = gnext[k];
#pragma unroll
            for (int jj = HR - 1; jj >= 0; --jj) {
                const int j = h * HR + jj;
                const f32x2 pv = *reinterpret_cast<const f32x2*>(&pt[j * XS_ROWP + 2 * lane]);
                f32x2 xw[W];
#pragma unroll
                for (int k = 0; k < W; ++k) {
                    xw[k] = xs_pair<T>::up(cur.x[j + k]);
                    if (j + k < W - 1) xw[k] = (l0 + j + k - (W - 1) < 0) ? (f32x2){0.f, 0.f} : xw[k];  // left zero padding (first tile of the sequence)
                }
                f32x2 pre = bias;
#pragma unroll
                for (int k = 0; k < W; ++k) pre += w[k] * xw[k];
                const f32x2 ex = pre * (-LOG2E);
                const f32x2 den = (f32x2){fast_exp2(ex.x), fast_exp2(ex.y)} + 1.0f;
                const f32x2 sg = {fast_rcp(den.x), fast_rcp(den.y)};
                const f32x2 fac = sg * (1.0f + pre * (1.0f - sg));                // silu'(pre)
                gv[jj] = (pv + xs_pair<T>::up(cur.du[j])) * fac;
#if DM_K4X_REPRO == 6                                                                 // the failing layout as hipcc compiles it
                const f32x2 gvo = gv[jj] * rw[j].own;
#elif DM_K4X_REPRO
                const f32x2 gvo = xs_mul_own_hi(gv[jj], *reinterpret_cast<const f32x2*>(&rw[j]));
#else
                const f32x2 gvo = gv[jj] * rw[j].own;                             // rows after the segment are someone else's
#endif
#pragma unroll
                for (int k = 0; k < W; ++k) dw[k] += gvo * xw[k];
                db += gvo;
            }
#pragma unroll
            for (int k = 0; k < W - 1; ++k) gnext[k] = gv[k];
            issue_du(more2 ? b_2 : bb, more2 ? dir_2 : 0, more2 ? t_2 : 0, more2, h, cur);      // this half's du registers are free
            // ---- dx[m] = sum_j w[j] * g[m + (W-1) - j], added to the token-order running sum (the rows of a direction are distinct tokens) ----
            uint32_t* ap[HR];
            uint32_t old[HR];
#pragma unroll
            for (int jj = HR - 1; jj >= 0; --jj) {
                const int j = h * HR + jj;
                ap[jj] = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(accs) + rw[j].acc_off) + lane;
                old[jj] = *ap[jj];
            }
#pragma unroll
            for (int jj = HR - 1; jj >= 0; --jj) {
                f32x2 dxv = w[W - 1] * gv[jj];
#pragma unroll
                for (int k = 0; k < W - 1; ++k) dxv += w[W - 2 - k] * gv[jj + 1 + k];
                const f32x2 o = xs_pair<T>::up(old[jj]) + dxv;
                *ap[jj] = xp_mfma<T>::pack(o.x, o.y);
            }
        }
        if (t == 0) {
            __syncthreads();                                                      // the next direction adds to rows other waves wrote
            if (dir == p.ndir - 1) {                                              // the sample is complete: dx leaves as 16-byte pieces, acc starts over
                const rsrc_t r_dx = make_rsrc((T*)p.dx + xs_uniform64((int64_t)bb * p.dx_ss + c0));
                const int sl_dx = (int)p.dx_sl * ES;
                const int piece = tid & 15;
                const xp_u32x4 z = {0u, 0u, 0u, 0u};
                for (int r = tid >> 4; r < L; r += XS_THREADS / 16) {
                    xp_u32x4* const q = reinterpret_cast<xp_u32x4*>(accs + r * AW + 4 * piece);
                    __builtin_amdgcn_raw_buffer_store_b128(*q, r_dx, piece * 16 + r * sl_dx, 0, 0);      // (the row differs inside a wave: lane offset)
                    *q = z;
                }
                __syncthreads();
            }
        }
    };

    __syncthreads();                                                              // acc zeroed, tables and fragments in place
    xs_bufs<T, RT> bufA, bufB;
    // the tile after (bb, dir, t)
    auto after = [&](int bb, int dir, int t, int& b2, int& d2, int& t2) {
        b2 = bb; d2 = dir; t2 = t - 1;
        if (t2 < 0) { t2 = nt - 1; d2 = dir + 1; }
        if (d2 == p.ndir) { d2 = 0; b2 = bb + bstep; }
    };
    int bb = b_first, dir = 0, t = nt - 1;
    {
        int b1, d1, t1;
        after(bb, dir, t, b1, d1, t1);
        issue(bb, dir, t, true, tokens(dir, t), bufA);
        issue_du(bb, dir, t, true, 1, bufA);
        issue_du(bb, dir, t, true, 0, bufA);
        const bool m1 = b1 < p.batch;
        issue_du(m1 ? b1 : bb, m1 ? d1 : 0, m1 ? t1 : 0, m1, 1, bufB);
        issue_du(m1 ? b1 : bb, m1 ? d1 : 0, m1 ? t1 : 0, m1, 0, bufB);
    }
    while (bb < p.batch) {
        int b1, d1, t1, b2, d2, t2, b3, d3, t3;
        after(bb, dir, t, b1, d1, t1);
        after(b1, d1, t1, b2, d2, t2);
        step(bb, dir, t, b1 < p.batch, b1, d1, t1, b2 < p.batch, b2, d2, t2, bufA, bufB);
        if (b1 >= p.batch) break;
        after(b2, d2, t2, b3, d3, t3);
        step(b1, d1, t1, b2 < p.batch, b2, d2, t2, b3 < p.batch, b3, d3, t3, bufB, bufA);
        bb = b2; dir = d2; t = t2;
    }
    {   // dw | db: the 8 segments' sums through LDS (the product tiles are free), into the partial row of the first sample; zeros into
        // the rows of the workgroup's other samples (the layout of the kernel above: one row per sample)
        f32x2* const red = reinterpret_cast<f32x2*>(&ptile[0][0]);               // [wave][5][64]
#pragma unroll
        for (int k = 0; k < W; ++k) red[(wave * (W + 1) + k) * WAVE + lane] = dw[k];
        red[(wave * (W + 1) + W) * WAVE + lane] = db;
        __syncthreads();
        if (tid < (W + 1) * WAVE) {
            const int k = tid >> 6, ln = tid & 63;
            f32x2 sum = red[k * WAVE + ln];
#pragma unroll
            for (int v = 1; v < XS_NW; ++v) sum += red[(v * (W + 1) + k) * WAVE + ln];
            const int ch = c0 + 2 * ln;
            const int b_end = (p.flags & DM_FLAG_PARTIAL_COMPACT) ? b_first + 1 : p.batch;      // compact: the caller sums the first rows only
            for (int b2 = b_first; b2 < b_end; b2 += bstep) {
                if (k < W) {
                    const int64_t row = (int64_t)b2 * (p.part_ss ? p.part_ss : (int64_t)p.dim * W);
                    p.dw_partial[row + (int64_t)ch * W + k] = sum.x;
                    p.dw_partial[row + (int64_t)(ch + 1) * W + k] = sum.y;
                } else if (p.db_partial) {
                    const int64_t rb = (int64_t)b2 * (p.part_ss ? p.part_ss : (int64_t)p.dim);
                    p.db_partial[rb + ch] = sum.x;
                    p.db_partial[rb + ch + 1] = sum.y;
                }
                sum = (f32x2){0.f, 0.f};
            }
        }
    }
}

static int xs_cu_count() {
    static int n = 0;
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// one workgroup per CU, in units of (8 XCDs x the slabs of a sample); fewer when the batch is small
static int xs_streams(const dm_conv_xproj_bwd_args& a) {
    int nstream = xs_cu_count() / 8 / (a.dim / XS_CS);
    if (nstream < 1) nstream = 1;
    if (nstream > (a.batch + 7) / 8) nstream = (a.batch + 7) / 8;
    return nstream;
}

// DM_K4X_SLAB=0 never, =1 whenever it is legal, unset: sequences of 32 .. 256 rows
static bool xs_use_slab(const dm_conv_xproj_bwd_args& a) {
    if (a.seqlen > XS_MAXL || a.dim % XS_CS || a.ndir > XS_MAXDIR) return false;
    if (((uintptr_t)a.dx & 15) || (a.dx_ss & 7) || (a.dx_sl & 7)) return false;   // 16-byte dx pieces
    const char* e = getenv("DM_K4X_SLAB");
    if (e && *e) return *e != '0';
    return a.seqlen >= 32;
}

template <typename T, typename TW, int W, int D>
static void launch_xpb(const dm_conv_xproj_bwd_args& a, hipStream_t st) {
    dim3 grid(a.ndir * a.batch), block(XP_THREADS);
    const bool silu = (a.flags & DM_FLAG_SILU) != 0;
    if constexpr (W == 4) {                                          // the merged form is built for the mixer's call pattern
        if ((a.flags & DM_FLAG_DX_MERGED) && silu && a.row_index) {
            if (xs_use_slab(a)) {
                const int rows = (a.seqlen + XS_NW - 1) / XS_NW + 3;                 // a wave's rows per direction
                const int nslab = a.dim / XS_CS, nstream = xs_streams(a);
                const dim3 g(8 * nslab * nstream), blk(XS_THREADS);
                if (14 * ((rows + 13) / 14) < 16 * ((rows + 15) / 16)) hipLaunchKernelGGL((conv_xproj_bwd_slab_kernel<T, TW, 14>), g, blk, 0, st, a);
                else hipLaunchKernelGGL((conv_xproj_bwd_slab_kernel<T, TW, 16>), g, blk, 0, st, a);
                return;
            }
            hipLaunchKernelGGL((conv_xproj_bwd_kernel<T, TW, W, true, D, true, true>), dim3(a.batch), block, 0, st, a);
            return;
        }
    }
    if (a.row_index) {
        if (silu) hipLaunchKernelGGL((conv_xproj_bwd_kernel<T, TW, W, true, D, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_xproj_bwd_kernel<T, TW, W, false, D, true>), grid, block, 0, st, a);
    } else {
        if (silu) hipLaunchKernelGGL((conv_xproj_bwd_kernel<T, TW, W, true, D, false>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((conv_xproj_bwd_kernel<T, TW, W, false, D, false>), grid, block, 0, st, a);
    }
}

template <typename T, typename TW, int W>
static int xpb_by_dim(const dm_conv_xproj_bwd_args& a, hipStream_t st) {
    switch (a.dim) {
        case 1024: launch_xpb<T, TW, W, 1024>(a, st); break;
#ifndef DM_FAST_BUILD
        case 512: launch_xpb<T, TW, W, 512>(a, st); break;
        case 256: launch_xpb<T, TW, W, 256>(a, st); break;
#endif
        case 128: launch_xpb<T, TW, W, 128>(a, st); break;
        default: set_error("dm_gather_conv1d_xproj_bwd: dim %d not instantiated (128, 256, 512, 1024)", a.dim); return DM_ERR_ARG;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gather_conv1d_xproj_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T, typename TW>
static int xpb_by_width(const dm_conv_xproj_bwd_args& a, hipStream_t st) {
    switch (a.width) {
        case 4: return xpb_by_dim<T, TW, 4>(a, st);
#ifndef DM_FAST_BUILD
        case 3: return xpb_by_dim<T, TW, 3>(a, st);
        case 2: return xpb_by_dim<T, TW, 2>(a, st);
#endif
        default: set_error("dm_gather_conv1d_xproj_bwd: width %d not in {2,3,4}", a.width); return DM_ERR_ARG;
    }
}

}  // namespace dm

extern "C" int dm_gather_conv1d_xproj_bwd_supported(int dim, int nproj, int io_dtype) {
    const bool d_ok = dim == 128 || dim == 256 || dim == 512 || dim == 1024;
    return (d_ok && nproj == 64 && (io_dtype == DM_BF16 || io_dtype == DM_F16)) ? 1 : 0;      // DiffMa: dt_rank 32 + 2 * d_state 16
}

extern "C" int dm_gather_conv1d_xproj_bwd_slab(const dm_conv_xproj_bwd_args* args, void*) {
    if (!args) return 0;
    const dm_conv_xproj_bwd_args& a = *args;
    if (!((a.flags & DM_FLAG_DX_MERGED) && (a.flags & DM_FLAG_SILU) && a.row_index && a.width == 4 && dm::xs_use_slab(a))) return 0;
    const int rows = 8 * dm::xs_streams(a);
    return rows < a.batch ? rows : a.batch;
}

extern "C" int dm_gather_conv1d_xproj_bwd(const dm_conv_xproj_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gather_conv1d_xproj_bwd: null args"); return DM_ERR_ARG; }
    const dm_conv_xproj_bwd_args& a = *args;
    if (!a.x || !a.weight || !a.du || !a.dxdbl || !a.wxt || !a.dx || !a.dw_partial) { set_error("dm_gather_conv1d_xproj_bwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ndir <= 0) { set_error("dm_gather_conv1d_xproj_bwd: non-positive size"); return DM_ERR_ARG; }
    if (a.ndir > 1 && !a.row_index) { set_error("dm_gather_conv1d_xproj_bwd: ndir>1 needs row_index"); return DM_ERR_ARG; }
    if ((a.flags & DM_FLAG_DX_MERGED) && !(a.width == 4 && (a.flags & DM_FLAG_SILU) && a.row_index)) {
        set_error("dm_gather_conv1d_xproj_bwd: DM_FLAG_DX_MERGED is built for width 4, SiLU, row-index tables"); return DM_ERR_ARG;
    }
    if ((a.flags & DM_FLAG_PARTIAL_COMPACT) && !dm_gather_conv1d_xproj_bwd_slab(args, nullptr)) {
        set_error("dm_gather_conv1d_xproj_bwd: DM_FLAG_PARTIAL_COMPACT is for launches dm_gather_conv1d_xproj_bwd_slab() accepts (it returned 0)");
        return DM_ERR_ARG;
    }
    if (!dm_gather_conv1d_xproj_bwd_supported(a.dim, a.nproj, a.io_dtype)) {
        set_error("dm_gather_conv1d_xproj_bwd: needs 16-bit I/O, dim in {128,256,512,1024}, nproj = 64 (got dim %d nproj %d dtype %d)", a.dim, a.nproj, a.io_dtype);
        return DM_ERR_ARG;
    }
    if (a.x_sd != 1 || a.du_sd != 1 || a.dx_sd != 1) { set_error("dm_gather_conv1d_xproj_bwd: needs token-major tensors"); return DM_ERR_LAYOUT; }
    const bool odd = ((uintptr_t)a.x & 3) || (a.x_sb & 1) || (a.x_sl & 1) || ((uintptr_t)a.du & 3) || (a.du_ss & 1) || (a.du_sl & 1) ||
                     ((uintptr_t)a.dx & 3) || (a.dx_ss & 1) || (a.dx_sl & 1) || ((uintptr_t)a.dxdbl & 15) || (a.xd_sr & 7) || ((uintptr_t)a.wxt & 15);
    if (odd) { set_error("dm_gather_conv1d_xproj_bwd: x / du / dx need 4-byte aligned rows, dxdbl and wxt 16-byte aligned rows"); return DM_ERR_LAYOUT; }
    hipStream_t st = (hipStream_t)stream;
    const bool wf32 = a.w_dtype == DM_F32;
    if (!wf32 && a.w_dtype != a.io_dtype) { set_error("dm_gather_conv1d_xproj_bwd: w_dtype must be fp32 or io_dtype"); return DM_ERR_DTYPE; }
    if (a.io_dtype == DM_BF16) return wf32 ? xpb_by_width<bf16_t, float>(a, st) : xpb_by_width<bf16_t, bf16_t>(a, st);
    return wf32 ? xpb_by_width<f16_t, float>(a, st) : xpb_by_width<f16_t, f16_t>(a, st);
}
