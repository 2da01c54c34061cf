// K6  dm_ssd_fwd -- Mamba-2 state-space duality, single chunk, on the matrix pipe (forward, no-grad path).
//
// Replaces the SSD core of mamba_split_conv1d_scan_combined (reference call block/mamba2.py:392-410; mathematics SURVEY.md A.2)
// after the conv: with chunk_size 256 >= L the operator is ONE chunk, and because Mamba-2's decay is a scalar per head the
// recurrence factorises into dense products (the "dual" quadratic form):
//     s_l   = A_h * cumsum(dt)_l                          (log-decay, <= 0 and decreasing)
//     G     = (C B^T) .* exp(s_l - s_i) [i <= l]          [L x L], K = d_state
//     Y     = (G diag(dt)) X + D_h X                      [L x P], K = L          out = Y * silu(z)
// One wave64 owns one (sequence, head): 32 x 32 tiles, v_mfma_f32_32x32x16_{bf16,f16}.  The transposed score tile
// G^T = (dt .* B)_it C_lt^T is produced with keys as rows and queries as columns, so its accumulator registers ARE the
// A-operand of the second product (row = query l, K slots = keys in the order 4*(lane>>5) + 8*r4 + r): the decay factor and
// the causal mask are applied in registers, the tile is rounded to 16 bit and fed straight back -- no LDS round trip, no
// cross-lane move.  X is held as B-operand fragments in that same key order for the whole sequence (112 VGPRs), so the inner
// loop has no memory operation at all except the broadcast reads of the log-decays from LDS.  z is gathered and the output
// scattered through the row-index tables exactly like the scan kernels (CrossScan / CrossMerge folded into addressing); the
// per-head dt is read in the kernel (token order, through the gather table), no [nseq, L, Din] delta tensor exists.
// 16-bit I/O only (the score tile is rounded to the I/O dtype); fp32 I/O and the training path stay on the A-shared scan.
#include <type_traits>
#include "dm_common.h"

namespace dm {

constexpr int SSD_TILE = 32;
constexpr int SSD_MAXT = 7;                       // L <= 224
constexpr int SSD_MAXL = SSD_TILE * SSD_MAXT;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ssd_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ssd_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ssd_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct ssd_ops;
template <> struct ssd_ops<bf16_t> {
    static __device__ __forceinline__ f32x16 mfma(const ssd_u32x4& a, const ssd_u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ssd_bf16x8, a), __builtin_bit_cast(ssd_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        uint32_t r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
    static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
};
template <> struct ssd_ops<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 mfma(const ssd_u32x4& a, const ssd_u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ssd_f16x8, a), __builtin_bit_cast(ssd_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
    static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(h2, w).x; }
    static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(h2, w).y; }
};

template <typename T>
__global__ __launch_bounds__(64) void ssd_fwd_kernel(const dm_ssd_fwd_args p) {
    using O = ssd_ops<T>;
    constexpr int ES = (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) float s2_lds[2][SSD_MAXL + 32];    // log2-domain log-decay (prefix sums, ping-pong)
    __shared__ __attribute__((aligned(16))) float dt_lds[SSD_MAXL];
    __shared__ int zi_lds[SSD_MAXL], oi_lds[SSD_MAXL];

    const int lane = threadIdx.x;
    const int col = lane & 31, kh = lane >> 5;
    const int h = blockIdx.x, s = blockIdx.y;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int nt_l = (L + SSD_TILE - 1) / SSD_TILE;
    const int32_t* __restrict__ zidx = p.z_row_index ? p.z_row_index + (int64_t)dir * L : nullptr;
    const int32_t* __restrict__ oidx = p.out_row_index ? p.out_row_index + (int64_t)dir * L : nullptr;
    const float Ah = p.A[h] * LOG2E, Dh = p.D ? p.D[h] : 0.0f, bias = p.dt_bias ? p.dt_bias[h] : 0.0f;
    const rsrc_t r_x = make_rsrc((const T*)p.x + (int64_t)s * p.x_ss + (int64_t)h * 64);
    const rsrc_t r_B = make_rsrc((const T*)p.B + (int64_t)s * p.B_ss);
    const rsrc_t r_C = make_rsrc((const T*)p.C + (int64_t)s * p.C_ss);
    const rsrc_t r_z = make_rsrc(p.z ? (const T*)p.z + (int64_t)sb * p.z_ss + (int64_t)h * 64 : nullptr);
    const rsrc_t r_o = make_rsrc((T*)p.out + (int64_t)s * p.o_ss + (int64_t)h * 64);
    const T* __restrict__ dtp = (const T*)p.dt + (int64_t)sb * p.dt_sb + h;
    const int sl_x = (int)p.x_sl * ES, sl_B = (int)p.B_sl * ES, sl_C = (int)p.C_sl * ES, sl_z = (int)p.z_sl * ES, sl_o = (int)p.o_sl * ES;

    // ---- per-position scalars: dt = softplus(raw + bias), log-decay prefix sums, row tables -------------------------------
#pragma unroll
    for (int k = 0; k < SSD_MAXL / WAVE + 1; ++k) {
        const int l = lane + WAVE * k;
        if (l < SSD_MAXL) {
            float dtv = 0.0f;
            int zr = 0, orow = 0;
            if (l < L) {
                zr = zidx ? zidx[l] : l;
                orow = oidx ? oidx[l] : l;
                dtv = softplus_f(io<T>::ld(dtp + (int64_t)zr * p.dt_sl) + bias);
            }
            dt_lds[l] = dtv;
            s2_lds[0][l] = Ah * dtv;
            zi_lds[l] = zr;
            oi_lds[l] = orow;
        }
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int off = 1; off < SSD_MAXL; off <<= 1) {                       // inclusive prefix sum (Hillis-Steele, one wave)
#pragma unroll
        for (int k = 0; k < SSD_MAXL / WAVE + 1; ++k) {
            const int l = lane + WAVE * k;
            if (l < SSD_MAXL) s2_lds[cur ^ 1][l] = s2_lds[cur][l] + (l >= off ? s2_lds[cur][l - off] : 0.0f);
        }
        __syncthreads();
        cur ^= 1;
    }
    const float* const s2 = s2_lds[cur];

    // ---- operands resident for the whole sequence ---------------------------------------------------------------------------
    // (dt .* B) rows as A-fragments of the score product: lane (row i = col, kh) holds dt_i * B[i][8kh .. 8kh+7]
    ssd_u32x4 bfrag[SSD_MAXT];
#pragma unroll
    for (int it = 0; it < SSD_MAXT; ++it) {
        const int i = SSD_TILE * it + col;
        const int ic = i < L ? i : L - 1;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(r_B, ic * sl_B + kh * 8 * ES, 0, 0);       // per-lane row: all of it in the VGPR offset
        const float dti = (i < L) ? dt_lds[ic] : 0.0f;                                                    // rows past the end: zero
#pragma unroll
        for (int w = 0; w < 4; ++w) bfrag[it][w] = O::pack(O::lo(q[w]) * dti, O::hi(q[w]) * dti);
    }
    // X as B-fragments of the output product, keys in accumulator order: slot e of (it, ks) is key 32it + 4kh + 8(2ks + e/4) + e%4
    ssd_u32x4 xfrag[SSD_MAXT][2][2];
#pragma unroll
    for (int it = 0; it < SSD_MAXT; ++it)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                uint32_t w4[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    uint32_t pr[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = 2 * e2 + q;
                        const int i = SSD_TILE * it + 4 * kh + 8 * (2 * ks + (e >> 2)) + (e & 3);
                        // (rows past the end are clamped and zeroed by a select: no exec-masked branch per load)
                        const int ic = i < L ? i : L - 1;
                        const uint32_t v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r_x, ic * sl_x + (32 * nt + col) * ES, 0, 0);
                        pr[q] = (i < L) ? v : 0u;
                    }
                    w4[e2] = pr[0] | (pr[1] << 16);
                }
                xfrag[it][ks][nt] = (ssd_u32x4){w4[0], w4[1], w4[2], w4[3]};
            }

    for (int lt = 0; lt < nt_l; ++lt) {
        const int lq = SSD_TILE * lt + col;                              // this lane's query in the score tile (a column of G^T)
        // C rows as the B-operand: lane (col l, kh) holds C[l][8kh .. +7]; queries past the end produce rows that are never stored
        const int lqc = lq < L ? lq : L - 1;
        const auto qc = __builtin_amdgcn_raw_buffer_load_b128(r_C, lqc * sl_C + kh * 8 * ES, 0, 0);
        const ssd_u32x4 cfrag = {qc[0], qc[1], qc[2], qc[3]};
        const float s2l = s2[lq];
        f32x16 yacc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[nt][r] = 0.0f;
#pragma unroll
        for (int it = 0; it < SSD_MAXT; ++it) {
            if (it <= lt) {                                               // wave-uniform
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = 0.0f;
                g = O::mfma(bfrag[it], cfrag, g);                         // G^T tile: rows = keys 32it + 4kh + 8r4 + r, columns = queries
                uint32_t wp[8];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const f32x4 si = *reinterpret_cast<const f32x4*>(&s2[SSD_TILE * it + 4 * kh + 8 * r4]);
                    float wv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = SSD_TILE * it + 4 * kh + 8 * r4 + r;
                        const float e = fast_exp2(s2l - si[r]);
                        wv[r] = (it < lt || i <= lq) ? g[4 * r4 + r] * e : 0.0f;      // causal mask only bites on the diagonal tile
                    }
                    wp[2 * r4] = O::pack(wv[0], wv[1]);
                    wp[2 * r4 + 1] = O::pack(wv[2], wv[3]);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const ssd_u32x4 wf = {wp[4 * ks], wp[4 * ks + 1], wp[4 * ks + 2], wp[4 * ks + 3]};
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) yacc[nt] = O::mfma(wf, xfrag[it][ks][nt], yacc[nt]);
                }
            }
        }
        // ---- epilogue of the query tile: + D x, gate, scatter.  yacc[nt][4 r4 + r] = Y[32lt + 4kh + 8 r4 + r][32nt + col] ----
#pragma unroll
        for (int it = 0; it < SSD_MAXT; ++it) {                           // (compile-time index into xfrag)
            if (it == lt) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int l = SSD_TILE * lt + 4 * kh + 8 * r4 + r;
                        if (l < L) {
                            const int zr = zi_lds[l], orow = oi_lds[l];
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                const int e = 4 * (r4 & 1) + r;
                                const uint32_t xw = xfrag[it][r4 >> 1][nt][e >> 1];
                                const float xv = (e & 1) ? O::hi(xw) : O::lo(xw);
                                float y = yacc[nt][4 * r4 + r] + Dh * xv;
                                if (p.z) y *= silu_f(bio<T>::ld(r_z, zr * sl_z + (32 * nt + col) * ES, 0));
                                bio<T>::st(r_o, orow * sl_o + (32 * nt + col) * ES, 0, y);
                            }
                        }
                    }
            }
        }
    }
}

}  // namespace dm

extern "C" int dm_ssd_fwd_supported(int seqlen, int headdim, int dstate, int io_dtype) {
    return (seqlen >= 1 && seqlen <= dm::SSD_MAXL && headdim == 64 && dstate == 16 && (io_dtype == DM_BF16 || io_dtype == DM_F16)) ? 1 : 0;
}

extern "C" int dm_ssd_fwd(const dm_ssd_fwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_ssd_fwd: null args"); return DM_ERR_ARG; }
    const dm_ssd_fwd_args& a = *args;
    if (!a.x || !a.B || !a.C || !a.dt || !a.A || !a.out) { set_error("dm_ssd_fwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.nseq <= 0 || a.nheads <= 0 || a.seqlen <= 0) { set_error("dm_ssd_fwd: non-positive size"); return DM_ERR_ARG; }
    if (!dm_ssd_fwd_supported(a.seqlen, a.headdim, a.dstate, a.io_dtype)) {
        set_error("dm_ssd_fwd: needs 16-bit I/O, headdim 64, d_state 16, seqlen <= %d (got L %d P %d N %d dtype %d)", SSD_MAXL, a.seqlen, a.headdim, a.dstate, a.io_dtype);
        return DM_ERR_ARG;
    }
    if (a.nseq > 65535) { set_error("dm_ssd_fwd: nseq %d > 65535", a.nseq); return DM_ERR_ARG; }
    if (a.batch_per_dir > 0 && a.nseq % a.batch_per_dir != 0) { set_error("dm_ssd_fwd: nseq %% batch_per_dir != 0"); return DM_ERR_ARG; }
    if ((a.z_row_index == nullptr) != (a.out_row_index == nullptr)) { set_error("dm_ssd_fwd: both row-index tables or neither"); return DM_ERR_ARG; }
    if (((uintptr_t)a.B & 15) || ((uintptr_t)a.C & 15) || (a.B_ss & 7) || (a.B_sl & 7) || (a.C_ss & 7) || (a.C_sl & 7)) {
        set_error("dm_ssd_fwd: B / C rows must be 16-byte aligned"); return DM_ERR_LAYOUT;
    }
    dim3 grid(a.nheads, a.nseq), block(WAVE);
    hipStream_t st = (hipStream_t)stream;
    if (a.io_dtype == DM_BF16) hipLaunchKernelGGL((ssd_fwd_kernel<bf16_t>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((ssd_fwd_kernel<f16_t>), grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_ssd_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
