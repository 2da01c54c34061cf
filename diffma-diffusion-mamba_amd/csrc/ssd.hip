// K6  dm_ssd_fwd -- Mamba-2 state-space duality, single chunk, on the matrix pipe (forward).
//
// Replaces the SSD core of mamba_split_conv1d_scan_combined (reference call block/mamba2.py:392-410; mathematics SURVEY.md A.2)
// after the conv: with chunk_size 256 >= L the operator is ONE chunk, and because Mamba-2's decay is a scalar per head the
// recurrence factorises into dense products (the "dual" quadratic form):
//     s_l   = A_h * cumsum(dt)_l                          (log-decay, <= 0 and decreasing)
//     G     = (C B^T) .* exp(s_l - s_i) [i <= l]          [L x L], K = d_state
//     Y     = (G diag(dt)) X + D_h X                      [L x P], K = L          out = Y * silu(z)
// One wave64 owns one (sequence, head, 32-column half): 32 x 32 tiles, v_mfma_f32_32x32x16_{bf16,f16}.  The transposed score tile
// G^T = (dt .* B)_it C_lt^T is produced with keys as rows and queries as columns, so its accumulator registers ARE the
// A-operand of the second product (row = query l, K slots = keys in the order 4*(lane>>5) + 8*r4 + r): the decay factor and
// the causal mask are applied in registers, the tile is rounded to 16 bit and fed straight back -- no LDS round trip, no
// cross-lane move.  X is held as B-operand fragments in that same key order for the whole sequence (56 VGPRs), so the inner
// loop has no memory operation at all except the broadcast reads of the log-decays from LDS.  z is gathered and the output
// scattered through the row-index tables exactly like the scan kernels (CrossScan / CrossMerge folded into addressing); the
// per-head dt is read in the kernel (token order, through the gather table), no [nseq, L, Din] delta tensor exists.
// 16-bit I/O only (the score tile is rounded to the I/O dtype); fp32 I/O stays on the A-shared scan.  Backward twin: ssd_bwd.hip.
#include <type_traits>
#include "dm_common.h"
#include "ssd_common.h"

namespace dm {

constexpr int SSD_TILE = 32;
constexpr int SSD_MAXT = 7;                       // L <= 224
constexpr int SSD_MAXL = SSD_TILE * SSD_MAXT;
constexpr int SSD_PITCH = 20;                     // dwords per staged tile row (32 channels = 16 dwords, +4 pad)

// One wave64 per (sequence, head, 32-column half of the head): 56 VGPRs of X fragments, ~150 registers, 3 waves per SIMD.
//
// Decay factorisation.  s2 = log2-domain log-decay prefix sums (decreasing), m_t = s2 just before tile t (m_0 = 0).  For a key
// tile it strictly before the query tile lt:   exp2(s2_l - s2_i) = alpha_l * delta(lt, it) * gamma_i   with
//     alpha_l = exp2(s2_l - m_lt),   gamma_i = exp2(m_{it+1} - s2_i),   delta = exp2(m_lt - m_{it+1}),     all three <= 1,
// so gamma (and dt) is folded into the B rows once, alpha into the C rows once per query tile, and an off-diagonal score tile
// costs one scalar exp and 16 multiplies.  Only the 7 diagonal tiles of 28 take the element-wise exp (on unscaled operands:
// there alpha * gamma could underflow while the true factor is O(1)).
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void ssd_fwd_kernel(const dm_ssd_fwd_args p) {
    using O = ssd_ops<T>;
    constexpr int ES = (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) float s2_lds[2][SSD_MAXL + 32];    // log2-domain log-decay (prefix sums, ping-pong)
    __shared__ __attribute__((aligned(16))) float dt_lds[SSD_MAXL];
    __shared__ int zi_lds[SSD_MAXL], oi_lds[SSD_MAXL];
    // 32-row x 32-channel staging tiles (row pitch 80 B: the two key halves of a fragment read land on disjoint banks): global
    // traffic is whole 16-byte pieces of a row, the (key, channel) element order of the fragments is produced by LDS reads
    __shared__ __attribute__((aligned(16))) uint32_t tile_a[SSD_TILE * SSD_PITCH], tile_b[SSD_TILE * SSD_PITCH];

    const int lane = threadIdx.x;
    const int col = lane & 31, kh = lane >> 5;
    const int trow = lane >> 1, thf = lane & 1;                          // staging role: row of the tile, 16-channel half of it
    const int h = blockIdx.x >> 1, half = blockIdx.x & 1, s = blockIdx.y;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int nt_l = (L + SSD_TILE - 1) / SSD_TILE;
    const int32_t* __restrict__ zidx = p.z_row_index ? p.z_row_index + (int64_t)dir * L : nullptr;
    const int32_t* __restrict__ oidx = p.out_row_index ? p.out_row_index + (int64_t)dir * L : nullptr;
    const float Ah = p.A[h] * LOG2E, Dh = p.D ? p.D[h] : 0.0f, bias = p.dt_bias ? p.dt_bias[h] : 0.0f;
    const rsrc_t r_x = make_rsrc((const T*)p.x + (int64_t)s * p.x_ss);
    const rsrc_t r_B = make_rsrc((const T*)p.B + (int64_t)s * p.B_ss);
    const rsrc_t r_C = make_rsrc((const T*)p.C + (int64_t)s * p.C_ss);
    const rsrc_t r_z = make_rsrc(p.z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_o = make_rsrc((T*)p.out + (int64_t)s * p.o_ss);
    const T* __restrict__ dtp = (const T*)p.dt + (int64_t)sb * p.dt_sb + h;
    const int sl_x = (int)p.x_sl * ES, sl_B = (int)p.B_sl * ES, sl_C = (int)p.C_sl * ES, sl_z = (int)p.z_sl * ES, sl_o = (int)p.o_sl * ES;

    // ---- every load that does not depend on the decays goes out first: one HBM round trip covers X, B and the dt gather ------
    const int cb = (h * 64 + half * 32 + thf * 16) * ES;                 // byte offset of this lane's 16 staged channels in a row
    ssd_u32x4 xq[SSD_MAXT][2], bq[SSD_MAXT];
#pragma unroll
    for (int it = 0; it < SSD_MAXT; ++it) {
        const int i = SSD_TILE * it + trow;
        const int ic = i < L ? i : L - 1;                                 // rows past the end: clamped here, zeroed below
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r_x, ic * sl_x + cb + 16 * q, 0, 0);
            xq[it][q] = (ssd_u32x4){v[0], v[1], v[2], v[3]};
        }
        const int j = SSD_TILE * it + col;
        const int jc = j < L ? j : L - 1;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r_B, jc * sl_B + kh * 8 * ES, 0, 0);
        bq[it] = (ssd_u32x4){v[0], v[1], v[2], v[3]};
    }

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
    auto m_of = [&](int t) -> float { return t == 0 ? 0.0f : s2[SSD_TILE * t - 1]; };   // log-decay just before tile t

    // ---- operands resident for the whole sequence ---------------------------------------------------------------------------
    // (gamma .* dt .* B) rows as A-fragments of the off-diagonal score products: lane (row i = col, kh) holds 8 states
    ssd_u32x4 bfrag[SSD_MAXT];
#pragma unroll
    for (int it = 0; it < SSD_MAXT; ++it) {
        const int i = SSD_TILE * it + col;
        const int ic = i < L ? i : L - 1;
        const float sc = (i < L) ? dt_lds[ic] * fast_exp2(m_of(it + 1 < SSD_MAXT ? it + 1 : it) - s2[ic]) : 0.0f;   // (the last tile is never off-diagonal)
#pragma unroll
        for (int w = 0; w < 4; ++w) bfrag[it][w] = O::pack(O::lo(bq[it][w]) * sc, O::hi(bq[it][w]) * sc);
    }
    // X as B-fragments of the output product, keys in accumulator order: slot e of (it, ks) is key 32it + 4kh + 8(2ks + e/4) + e%4
    ssd_u32x4 xfrag[SSD_MAXT][2];
    {
        const uint16_t* const ta16 = reinterpret_cast<const uint16_t*>(tile_a);
#pragma unroll
        for (int it = 0; it < SSD_MAXT; ++it) {
            const bool live = SSD_TILE * it + trow < L;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<ssd_u32x4*>(&tile_a[trow * SSD_PITCH + thf * 8 + 4 * q]) = live ? xq[it][q] : (ssd_u32x4){0u, 0u, 0u, 0u};
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint32_t w4[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const int k0 = 4 * kh + 8 * (2 * ks + (e2 >> 1)) + 2 * (e2 & 1);
                    const uint32_t lo = ta16[k0 * (2 * SSD_PITCH) + col], hi = ta16[(k0 + 1) * (2 * SSD_PITCH) + col];
                    w4[e2] = lo | (hi << 16);
                }
                xfrag[it][ks] = (ssd_u32x4){w4[0], w4[1], w4[2], w4[3]};
            }
            __syncthreads();
        }
    }

    // C rows (the B-operand of the score product: lane (col l, kh) holds C[l][8kh .. +7]) and the unscaled B rows of the diagonal
    // tile are fetched one query tile ahead
    auto load_cb = [&](int t, ssd_u32x4& c, ssd_u32x4& b) {
        const int j = SSD_TILE * t + col;
        const int jc = j < L ? j : L - 1;
        const auto vc = __builtin_amdgcn_raw_buffer_load_b128(r_C, jc * sl_C + kh * 8 * ES, 0, 0);
        const auto vb = __builtin_amdgcn_raw_buffer_load_b128(r_B, jc * sl_B + kh * 8 * ES, 0, 0);
        c = (ssd_u32x4){vc[0], vc[1], vc[2], vc[3]};
        b = (ssd_u32x4){vb[0], vb[1], vb[2], vb[3]};
    };
    ssd_u32x4 cnext, bnext;
    load_cb(0, cnext, bnext);
#pragma unroll
    for (int lt = 0; lt < SSD_MAXT; ++lt) {                              // (fully unrolled: every fragment index is a compile-time constant)
        if (lt >= nt_l) break;
        const int lq = SSD_TILE * lt + col;                              // this lane's query in the score tile (a column of G^T)
        const int lqc = lq < L ? lq : L - 1;
        const float s2l = s2[lqc], mlt = m_of(lt);
        const ssd_u32x4 craw = cnext, qb = bnext;                         // raw C for the diagonal tile, alpha-scaled for the others
        const ssd_u32x4 qc = craw;
        if (lt + 1 < SSD_MAXT) load_cb(lt + 1, cnext, bnext);
        ssd_u32x4 zq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};                // the gate tile of this query tile, in flight under the products
        const int lrow = SSD_TILE * lt + trow;
        const int lrc = lrow < L ? lrow : L - 1;
        if (p.z) {
            const int zr = zi_lds[lrc];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(r_z, zr * sl_z + cb + 16 * q, 0, 0);
                zq[q] = (ssd_u32x4){v[0], v[1], v[2], v[3]};
            }
        }
        const float alpha = fast_exp2(s2l - mlt);
        ssd_u32x4 cfrag;
#pragma unroll
        for (int w = 0; w < 4; ++w) cfrag[w] = O::pack(O::lo(qc[w]) * alpha, O::hi(qc[w]) * alpha);
        f32x16 yacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[r] = 0.0f;
        // ---- key tiles strictly before the query tile: factorised decay ----
#pragma unroll
        for (int it = 0; it < SSD_MAXT - 1; ++it) {
            if (it < lt) {                                                // wave-uniform
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = 0.0f;
                g = O::mfma(bfrag[it], cfrag, g);                         // rows = keys 32it + 4kh + 8r4 + r, columns = queries
                const float delta = fast_exp2(mlt - m_of(it + 1));
                uint32_t wp[8];
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const f32x2 gs = (f32x2){g[2 * r2], g[2 * r2 + 1]} * (f32x2){delta, delta};        // v_pk_mul_f32
                    wp[r2] = O::pack(gs.x, gs.y);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const ssd_u32x4 wf = {wp[4 * ks], wp[4 * ks + 1], wp[4 * ks + 2], wp[4 * ks + 3]};
                    yacc = O::mfma(wf, xfrag[it][ks], yacc);
                }
            }
        }
        // ---- the diagonal tile: element-wise decay and causal mask on unscaled operands ----
        {
            const int i = SSD_TILE * lt + col;
            const int ic = i < L ? i : L - 1;
            const float dti = (i < L) ? dt_lds[ic] : 0.0f;
            ssd_u32x4 bd;
#pragma unroll
            for (int w = 0; w < 4; ++w) bd[w] = O::pack(O::lo(qb[w]) * dti, O::hi(qb[w]) * dti);
            f32x16 g;
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = 0.0f;
            g = O::mfma(bd, craw, g);
            uint32_t wp[8];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 si = *reinterpret_cast<const f32x4*>(&s2[SSD_TILE * lt + 4 * kh + 8 * r4]);
                float wv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ik = SSD_TILE * lt + 4 * kh + 8 * r4 + r;
                    wv[r] = (ik <= lq) ? g[4 * r4 + r] * fast_exp2(s2l - si[r]) : 0.0f;
                }
                wp[2 * r4] = O::pack(wv[0], wv[1]);
                wp[2 * r4 + 1] = O::pack(wv[2], wv[3]);
            }
#pragma unroll
            for (int it = 0; it < SSD_MAXT; ++it) {                       // (compile-time index into xfrag)
                if (it == lt) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const ssd_u32x4 wf = {wp[4 * ks], wp[4 * ks + 1], wp[4 * ks + 2], wp[4 * ks + 3]};
                        yacc = O::mfma(wf, xfrag[it][ks], yacc);
                    }
                }
            }
        }
        // ---- epilogue of the query tile: + D x, gate, scatter.  yacc[4 r4 + r] = Y[32lt + 4kh + 8 r4 + r][this lane's channel] ----
        if (p.z) {
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<ssd_u32x4*>(&tile_a[trow * SSD_PITCH + thf * 8 + 4 * q]) = zq[q];
        }
        __syncthreads();
        {
            const uint16_t* const ta16 = reinterpret_cast<const uint16_t*>(tile_a);
            uint16_t* const tb16 = reinterpret_cast<uint16_t*>(tile_b);
#pragma unroll
            for (int it = 0; it < SSD_MAXT; ++it) {                       // (compile-time index into xfrag)
                if (it == lt) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 4 * kh + 8 * r4 + r;
                            const int e = 4 * (r4 & 1) + r;
                            const uint32_t xw = xfrag[it][r4 >> 1][e >> 1];
                            const float xv = (e & 1) ? O::hi(xw) : O::lo(xw);
                            float y = yacc[4 * r4 + r] + Dh * xv;
                            if (p.z) y *= silu_f(O::lo((uint32_t)ta16[row * (2 * SSD_PITCH) + col]));
                            tb16[row * (2 * SSD_PITCH) + col] = (uint16_t)(O::pack(y, 0.0f) & 0xffffu);
                        }
                }
            }
        }
        __syncthreads();
        if (lrow < L) {
            const int orow = oi_lds[lrow];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const ssd_u32x4 v = *reinterpret_cast<const ssd_u32x4*>(&tile_b[trow * SSD_PITCH + thf * 8 + 4 * q]);
                __builtin_amdgcn_raw_buffer_store_b128(v, r_o, orow * sl_o + cb + 16 * q, 0, 0);
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
    if (((uintptr_t)a.x & 15) || ((uintptr_t)a.out & 15) || (a.x_ss & 7) || (a.x_sl & 7) || (a.o_ss & 7) || (a.o_sl & 7) ||
        (a.z && (((uintptr_t)a.z & 15) || (a.z_ss & 7) || (a.z_sl & 7)))) {
        set_error("dm_ssd_fwd: x / z / out rows must be 16-byte aligned (tiles move as 16-byte pieces)"); return DM_ERR_LAYOUT;
    }
    dim3 grid(a.nheads * 2, a.nseq), block(WAVE);          // two 32-column halves per head
    hipStream_t st = (hipStream_t)stream;
    if (a.io_dtype == DM_BF16) hipLaunchKernelGGL((ssd_fwd_kernel<bf16_t>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((ssd_fwd_kernel<f16_t>), grid, block, 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_ssd_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
