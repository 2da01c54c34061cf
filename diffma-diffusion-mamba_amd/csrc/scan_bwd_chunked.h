#pragma once
// K2c  chunk-parallel selective-scan BACKWARD for SMALL launches (graphed small-batch training: the reference's own
// config/brain.yaml trains at global batch 8, i.e. 1-8 samples per GPU).
//
// The sequential kernel (scan_bwd_impl.h) puts one wave on (sequence, 64 channels) for all L steps; at nseq 24 that is 96
// workgroups on 256 CUs and every launch costs the full 196-step dependent chain (~190 us, a quarter of the graphed batch-8
// step).  The adjoint recurrence is LINEAR in its carry,
//       G_j = C_j * gy_j + carry_{j+1},      carry_j = a_j * G_j            (a_j = exp(delta_j * A), reverse time)
// so the time axis is cut into NW chunks of LC steps, one WAVE per chunk inside a workgroup that owns (sequence, 64 channels):
//   pass 1   every wave sweeps its chunk backwards from carry = 0 with the bare recursion (no states, no outputs) and
//            publishes, per state, the carry leaving the chunk on the left  c_loc  and the chunk's total decay
//            Q = exp(A * sum(delta))                                                                          -> LDS
//   combine  after one barrier wave c folds the chunks to its right:  X <- c_loc(j) + Q(j) * X,  j = NW-1 .. c+1
//   pass 2   every wave runs the full backward of its chunk (recompute from the forward's checkpoints -- chunk starts are
//            multiples of the checkpoint spacing --, adjoint sweep, all gradients) starting from the true carry X.
// The dependent chain is ~(0.25 + 1) * L/NW steps instead of L.  dB/dC: a wave owns its steps alone, so the lane sum is
// finished in the wave (matrix-pipe lane-group sums as in the sequential kernel, then a DPP row sum) and stored as ONE partial
// row per (step, 64 channels); dA/dD/dbias of the NW waves meet in LDS.  This is the "wavefront-parallel" form of the north
// star taken where latency, not throughput, is the limit; selection by launch size: dm_scan_bwd_chunked().
#include <cstdlib>
#include "scan_bwd_impl.h"

namespace dm {

// sum over the 16 lanes of a DPP row (every lane of the row gets the total)
__device__ __forceinline__ float row_sum16(float x) {
#define DM_ROW_ADD(CTRL) x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xF, 0xF, true))
    DM_ROW_ADD(0xB1);         // quad_perm [1,0,3,2]
    DM_ROW_ADD(0x4E);         // quad_perm [2,3,0,1]
    DM_ROW_ADD(0x141);        // row_half_mirror
    DM_ROW_ADD(0x140);        // row_mirror
#undef DM_ROW_ADD
    return x;
}

template <typename T, typename TBC, bool HAS_Z, bool IDX, int DMODE, int NW, int LC, bool ASH = false>   // DMODE: scan_bwd_impl.h
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2))) void scan_bwd_chunked_kernel(const mix_args<dm_scan_bwd_args> pm) {
    const dm_scan_bwd_args& p = pm.a[blockIdx.z];      // grid.z = congruent launches sharing this one (the two mixers of a block)
    constexpr int N = 16, NPL = N / 2, SUB = BWD_SUB, M = 2 * N, NSUB = LC / SUB;
    constexpr int ES = (int)sizeof(T);
    constexpr bool MFMA_RED = std::is_same<T, bf16_t>::value;
    static_assert(LC % SUB == 0, "chunks must start on checkpoints");
    __shared__ __attribute__((aligned(16))) float bc_lds[NW][LC][2 * N];        // [B row | C row] of every step of the wave's chunk
    __shared__ float xch_lds[NW][2][N + 2][WAVE];                                 // pass 1: [chunk][c_loc | Q][state][lane]; end: dA|dD|dbias partials

    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);               // wave = chunk index
    const int d_raw = blockIdx.x * WAVE + lane;
    const bool active = d_raw < p.dim;
    const int d = active ? d_raw : p.dim - 1;
    const int s = blockIdx.y;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int grp = (blockIdx.x * WAVE) / (p.dim / p.ngroups);
    const int nwg = gridDim.x;
    const int l0 = c * LC;
    const int nl = (L - l0 < LC) ? ((L - l0 > 0) ? L - l0 : 0) : LC;              // steps of this chunk (0 for waves past the end)

    const rsrc_t r_u = make_rsrc((const T*)p.u + (int64_t)s * p.u_ss);
    const rsrc_t r_dt = make_rsrc((const T*)p.delta + (int64_t)s * p.dt_ss);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_g = make_rsrc((const T*)p.dout + (int64_t)((IDX && !(p.flags & DM_FLAG_DOUT_PER_SEQ)) ? sb : s) * p.do_ss);
    const rsrc_t r_du = make_rsrc((T*)p.du + (int64_t)s * p.du_ss);
    const rsrc_t r_ddt = make_rsrc((T*)p.ddelta + (int64_t)s * p.ddt_ss);
    const rsrc_t r_dz = make_rsrc(HAS_Z ? (T*)p.dz + (int64_t)s * p.dz_ss : nullptr);
    const TBC* __restrict__ Bg = (const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg;
    const TBC* __restrict__ Cg = (const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg;
    constexpr bool CK_PACKED = std::is_same<T, bf16_t>::value;
    constexpr int CK_ROWS = CK_PACKED ? N / 2 : N;
    constexpr int H0W = CK_PACKED ? NPL : 2 * NPL;
    const int nck = (L + SUB - 1) / SUB;
    const rsrc_t r_ck = make_rsrc((const uint32_t*)p.ckpt + (int64_t)s * nck * CK_ROWS * p.dim);
    const int vo = d * ES, vo_ck = d * 4;
    const int sl_u = (int)p.u_sl * ES, sl_dt = (int)p.dt_sl * ES, sl_z = (int)p.z_sl * ES, sl_g = (int)p.do_sl * ES;
    const int sl_du = (int)p.du_sl * ES, sl_ddt = (int)p.ddt_sl * ES, sl_dz = (int)p.dz_sl * ES;
    const cptr<int32_t> zidx = IDX ? as_const(p.z_row_index + (int64_t)dir * L) : nullptr;
    const cptr<int32_t> oidx = IDX ? as_const(p.out_row_index + (int64_t)dir * L) : nullptr;

    // B/C rows of the chunk -> wave-private LDS slab
    for (int e = lane; e < LC * 2 * N; e += WAVE) {
        const int j = e / (2 * N), cc = e % (2 * N);
        const int l = (l0 + j < L) ? l0 + j : L - 1;
        bc_lds[c][j][cc] = (cc < N) ? io<TBC>::ld(Bg + (int64_t)l * p.B_sl + cc) : io<TBC>::ld(Cg + (int64_t)l * p.C_sl + cc - N);
    }
    f32x2 A2[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        A2[k].x = p.A[(int64_t)d * N + 2 * k] * LOG2E;
        A2[k].y = p.A[(int64_t)d * N + 2 * k + 1] * LOG2E;
    }
    const float Dv = p.D ? p.D[d] : 0.0f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.0f;

    // inputs of one sub-chunk (SUB steps); rows past the end re-read row L-1 and only ever meet g = 0
    struct Sub { float uu[SUB], dl[SUB], zz[SUB], gg[SUB]; int zrow[SUB]; };
    auto load_sub = [&](int sc, Sub& in, bool with_u) {
#pragma unroll
        for (int i = 0; i < SUB; ++i) {
            const int lr = l0 + sc * SUB + i;
            const int l = (lr < L) ? lr : L - 1;
            in.zrow[i] = IDX ? zidx[l] : l;
            const int orow = IDX ? oidx[l] : l;
            in.uu[i] = with_u ? bio<T>::ld(r_u, vo, l * sl_u) : 0.f;
            in.dl[i] = bio<T>::ld(r_dt, vo, l * sl_dt);
            in.zz[i] = HAS_Z ? bio<T>::ld(r_z, vo, in.zrow[i] * sl_z) : 0.f;
            in.gg[i] = bio<T>::ld(r_g, vo, orow * sl_g);
        }
    };
    auto finish_sub = [&](int sc, Sub& in) {                                      // softplus, zero gradients past the end
#pragma unroll
        for (int i = 0; i < SUB; ++i) {
            float x = in.dl[i];
            if (DMODE != 2) x += bias;
            if (DMODE == 1) x = softplus_f(x);
            in.dl[i] = x;
            in.gg[i] = ((l0 + sc * SUB + i) < L && active) ? in.gg[i] : 0.f;
        }
    };
    const int nsub = (nl + SUB - 1) / SUB;                                        // wave-uniform

    // ---- pass 1: bare adjoint recursion from carry = 0; total decay of the chunk -----------------------------------------
    f32x2 carry[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) carry[k] = (f32x2){0.f, 0.f};
    float sd = 0.f;
    for (int sc = nsub - 1; sc >= 0; --sc) {
        Sub in;
        load_sub(sc, in, false);
        finish_sub(sc, in);
#pragma unroll
        for (int i = SUB - 1; i >= 0; --i) {
            const bool valid = (l0 + sc * SUB + i) < L;                           // wave-uniform
            const float dlo = valid ? in.dl[i] : 0.f;                              // steps past the end: decay 1, no contribution
            sd += dlo;
            float gy = in.gg[i];
            if (HAS_Z) gy *= in.zz[i] * sigmoid_f(in.zz[i]);
            const float* crow = &bc_lds[c][sc * SUB + i][N];
            float a_sh = 0.f;
            if (ASH) a_sh = fast_exp2(A2[0].x * dlo);
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                f32x2 a, cc;
                if (ASH) {
                    a = (f32x2){a_sh, a_sh};
                } else {
                    const f32x2 t = A2[k] * dlo;
                    a.x = fast_exp2(t.x);
                    a.y = fast_exp2(t.y);
                }
                cc.x = crow[2 * k];
                cc.y = crow[2 * k + 1];
                carry[k] = a * (cc * gy + carry[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const f32x2 t = A2[k] * sd;
        xch_lds[c][0][2 * k][lane] = carry[k].x;
        xch_lds[c][0][2 * k + 1][lane] = carry[k].y;
        xch_lds[c][1][2 * k][lane] = fast_exp2(t.x);
        xch_lds[c][1][2 * k + 1][lane] = fast_exp2(t.y);
    }
    __syncthreads();
    // ---- combine: the carry entering this chunk from the right --------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NPL; ++k) carry[k] = (f32x2){0.f, 0.f};
    for (int j = NW - 1; j > c; --j) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            f32x2 cl, q;
            cl.x = xch_lds[j][0][2 * k][lane];
            cl.y = xch_lds[j][0][2 * k + 1][lane];
            q.x = xch_lds[j][1][2 * k][lane];
            q.y = xch_lds[j][1][2 * k + 1][lane];
            carry[k] = q * carry[k] + cl;
        }
    }
    __syncthreads();                                                              // xch_lds is reused for the parameter-gradient partials

    // ---- pass 2: the full backward of the chunk from the true carry ----------------------------------------------------------
    f32x2 dA[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) dA[k] = (f32x2){0.f, 0.f};
    float dD_acc = 0.f, dbias_acc = 0.f;
    u32x4_t sel_lo, sel_hi;
    if (MFMA_RED) mfma_selectors(lane, sel_lo, sel_hi);

    auto load_state = [&](int ci, uint32_t(&w)[H0W]) {                            // ci = global sub-chunk index (K2's convention)
        const int slot = (ci > 0 && ci * SUB < L) ? ci : ((ci > 0 && ci * SUB == L) ? 0 : -1);
        if (slot < 0) {
#pragma unroll
            for (int k = 0; k < H0W; ++k) w[k] = 0u;
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                if constexpr (CK_PACKED) {
                    if (k % 4 == 0) {                            // [chunk][N/8][d][4 words]: 16 bytes at a time
                        const u32x4_t qv = __builtin_amdgcn_raw_buffer_load_b128(r_ck, vo_ck * 4, (slot * (N / 8) + k / 4) * p.dim * 16, 0);
                        w[k] = qv[0]; w[k + 1] = qv[1]; w[k + 2] = qv[2]; w[k + 3] = qv[3];
                    }
                } else {
                    w[2 * k] = __builtin_amdgcn_raw_buffer_load_b32(r_ck, vo_ck, ((slot * N + 2 * k) * p.dim) * 4, 0);
                    w[2 * k + 1] = __builtin_amdgcn_raw_buffer_load_b32(r_ck, vo_ck, ((slot * N + 2 * k + 1) * p.dim) * 4, 0);
                }
            }
        }
    };
    auto unpack_state = [&](f32x2(&h)[NPL], const uint32_t(&w)[H0W]) {
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            if constexpr (CK_PACKED) {
                h[k].x = __uint_as_float(w[k] << 16);
                h[k].y = __uint_as_float(w[k] & 0xffff0000u);
            } else {
                h[k].x = __uint_as_float(w[2 * k]);
                h[k].y = __uint_as_float(w[2 * k + 1]);
            }
        }
    };

    for (int sc = nsub - 1; sc >= 0; --sc) {
        const int ci = l0 / SUB + sc;
        Sub in;
        load_sub(sc, in, true);
        uint32_t w0[H0W], w1[H0W];
        load_state(ci, w0);                                                       // state entering the sub-chunk
        load_state(ci + 1, w1);                                                   // state after its last step (= the next checkpoint, or slot 0)
        finish_sub(sc, in);
        f32x2 h[NPL], hs[SUB][NPL];
        unpack_state(h, w0);
#pragma unroll
        for (int i = 0; i < SUB; ++i) {                                           // recompute: hs[i] = state before step sc*SUB + i
#pragma unroll
            for (int k = 0; k < NPL; ++k) hs[i][k] = h[k];
            if (i < SUB - 1) {
                const float* brow = &bc_lds[c][sc * SUB + i][0];
                const float dlo = in.dl[i];
                const float du = dlo * in.uu[i];
                float a_sh = 0.f;
                if (ASH) a_sh = fast_exp2(A2[0].x * dlo);
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    f32x2 a, bb;
                    if (ASH) {
                        a = (f32x2){a_sh, a_sh};
                    } else {
                        const f32x2 t = A2[k] * dlo;
                        a.x = fast_exp2(t.x);
                        a.y = fast_exp2(t.y);
                    }
                    bb.x = brow[2 * k];
                    bb.y = brow[2 * k + 1];
                    h[k] = a * h[k] + bb * du;
                }
            }
        }
        unpack_state(h, w1);
        // a partial last sub-chunk (L not a multiple of SUB): the state after its last VALID step is slot 0; the steps past the
        // end are exact no-ops (g = 0), and the states "before" them are never multiplied with a non-zero gradient
#pragma unroll
        for (int i = SUB - 1; i >= 0; --i) {
            const int lraw = l0 + sc * SUB + i;
            const bool valid = lraw < L;                                          // wave-uniform
            const int l = valid ? lraw : L - 1;
            const float* brow = &bc_lds[c][sc * SUB + i][0];
            const float g = in.gg[i];
            float sz = 1.f, gy = g;
            if (HAS_Z) {
                sz = sigmoid_f(in.zz[i]);
                gy = g * in.zz[i] * sz;
            }
            const float dlo = in.dl[i];
            const float du = dlo * in.uu[i];
            float a_rev = 0.f;
            if (ASH) a_rev = fast_exp2(A2[0].x * dlo);
            f32x2 yp2 = (f32x2){0.f, 0.f}, GB2 = (f32x2){0.f, 0.f}, dlA2 = (f32x2){0.f, 0.f};
            float red[M];
            uint32_t pk_all[M / 2];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                f32x2 bb, cc, a;
                bb.x = brow[2 * k]; bb.y = brow[2 * k + 1];
                cc.x = brow[N + 2 * k]; cc.y = brow[N + 2 * k + 1];
                if (ASH) {
                    a = (f32x2){a_rev, a_rev};
                } else {
                    const f32x2 t = A2[k] * dlo;
                    a.x = fast_exp2(t.x);
                    a.y = fast_exp2(t.y);
                }
                const f32x2 hj = h[k];
                const f32x2 hp = hs[i][k];
                yp2 += cc * hj;
                const f32x2 G = cc * gy + carry[k];
                const f32x2 dCp = hj * gy;
                carry[k] = a * G;                                                 // (steps past the end: g = 0 and carry = 0, exact no-ops)
                const f32x2 Gt = carry[k] * hp;
                dlA2 += A2[k] * Gt;
                dA[k] += Gt * dlo;
                GB2 += G * bb;
                const f32x2 dBp = G * du;
                if constexpr (MFMA_RED) {
                    pk_all[k] = pack_bf16(dBp.x, dBp.y);
                    pk_all[NPL + k] = pack_bf16(dCp.x, dCp.y);
                } else {
                    red[2 * k] = dBp.x;
                    red[2 * k + 1] = dBp.y;
                    red[N + 2 * k] = dCp.x;
                    red[N + 2 * k + 1] = dCp.y;
                }
                h[k] = hp;
            }
            const float ypre = yp2.x + yp2.y + Dv * in.uu[i];
            const float GB = GB2.x + GB2.y;
            const float dlA = dlA2.x + dlA2.y;
            float ddl = in.uu[i] * GB + LN2 * dlA;
            const float duv = dlo * GB + gy * Dv;
            if (DMODE != 0) ddl *= (1.0f - fast_exp2(-dlo * LOG2E));
            dD_acc += gy * in.uu[i];
            dbias_acc += ddl;
            if (valid && active) {
                bio<T>::st(r_du, vo, l * sl_du, duv);
                bio<T>::st(r_ddt, vo, l * sl_ddt, ddl);
                if (HAS_Z) {
                    const float dzv = g * ypre * sz * (1.0f + in.zz[i] * (1.0f - sz));
                    bio<T>::st(r_dz, vo, in.zrow[i] * sl_dz, dzv);
                }
            }
            // ---- dB/dC of this step: finish the 64-lane sum inside the wave, one partial row per (step, 64 channels) ----
            float* const prow = p.dBC_partial + (((int64_t)s * L + l) * nwg + blockIdx.x) * (2 * N);
            if constexpr (MFMA_RED) {
#pragma unroll
                for (int g16 = 0; g16 < M / 16; ++g16) {                          // g16 = 0: dB, 1: dC;  register r of lane l = value 4*(l>>4) + r
                    const u32x4_t lo = {pk_all[8 * g16], pk_all[8 * g16 + 1], pk_all[8 * g16 + 2], pk_all[8 * g16 + 3]};
                    const u32x4_t hi = {pk_all[8 * g16 + 4], pk_all[8 * g16 + 5], pk_all[8 * g16 + 6], pk_all[8 * g16 + 7]};
                    f32x4 dsum = mfma_group_sum16(sel_lo, sel_hi, lo, hi);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsum[r] = row_sum16(dsum[r]);
                    if (valid && (lane & 15) == 0) *reinterpret_cast<f32x4*>(prow + g16 * N + 4 * (lane >> 4)) = dsum;
                }
            } else {
                lane_group_reduce<M>(red);                                        // register i = value 4*i + 2*b4 + b5
#pragma unroll
                for (int i8 = 0; i8 < M / 4; ++i8) {
                    const float tot = row_sum16(red[i8]);
                    if (valid && (lane & 15) == 0) prow[4 * i8 + 2 * ((lane >> 4) & 1) + (lane >> 5)] = tot;
                }
            }
        }
    }
    // ---- dA / dD / dbias: the NW waves of a (sequence, 64 channels) meet in LDS ------------------------------------------------
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        xch_lds[c][0][2 * k][lane] = dA[k].x;
        xch_lds[c][0][2 * k + 1][lane] = dA[k].y;
    }
    xch_lds[c][0][N][lane] = dD_acc;
    xch_lds[c][0][N + 1][lane] = dbias_acc;
    __syncthreads();
    if (c == 0 && active) {
        const int64_t pss_a = p.part_ss ? p.part_ss : (int64_t)p.dim * N, pss_d = p.part_ss ? p.part_ss : (int64_t)p.dim;
        float* dAp = p.dA_partial + (int64_t)s * pss_a + (int64_t)d * N;
#pragma unroll
        for (int n = 0; n < N + 2; ++n) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) acc += xch_lds[w][0][n][lane];
            if (n < N) dAp[n] = acc;
            else if (n == N) { if (p.dD_partial) p.dD_partial[(int64_t)s * pss_d + d] = acc; }
            else { if (p.dbias_partial) p.dbias_partial[(int64_t)s * pss_d + d] = acc; }
        }
    }
}

// Launch-size rule shared by the kernel dispatch and the host (which sizes the dB/dC partial rows: one per 64 channels here,
// one per 256 for the sequential kernel).  Measured break-even is near 250 workgroup-waves; the LDS footprint allows one
// workgroup per CU.
constexpr int BWD_CHUNKED_NW = 7, BWD_CHUNKED_LC_LONG = 28, BWD_CHUNKED_LC_SHORT = 8;
static inline int bwd_chunked_lc(int nseq, int dim, int seqlen, int dstate, int flags) {
    static const int env = [] { const char* e = getenv("DM_SCAN_BWD_CHUNKED"); return e ? atoi(e) : -1; }();   // 0 / 1: developer override
    const int forced = (flags & DM_FLAG_SCAN_SEQUENTIAL) ? 0 : ((flags & DM_FLAG_SCAN_CHUNKED) ? 1 : env);
    if (forced == 0 || dstate != 16) return 0;
    const int64_t waves = (int64_t)nseq * ((dim + WAVE - 1) / WAVE);
    if (!(waves <= 512 || forced == 1)) return 0;
    if (seqlen > 2 * BWD_CHUNKED_LC_LONG && seqlen <= BWD_CHUNKED_NW * BWD_CHUNKED_LC_LONG) return BWD_CHUNKED_LC_LONG;
    if (seqlen > 2 * BWD_CHUNKED_LC_SHORT && seqlen <= BWD_CHUNKED_NW * BWD_CHUNKED_LC_SHORT) return BWD_CHUNKED_LC_SHORT;
    return 0;
}

template <typename T, typename TBC, bool HAS_Z, bool IDX, int LC>
static void launch_bwd_chunked3(const dm_scan_bwd_args& a, hipStream_t st) {
    unsigned gz;
    const mix_args<dm_scan_bwd_args> m = mix_make(a, gz);
    dim3 grid((a.dim + WAVE - 1) / WAVE, a.nseq, gz), block(WAVE * BWD_CHUNKED_NW);
    const bool sp = (a.flags & DM_FLAG_DELTA_SOFTPLUS) != 0;
    if constexpr (HAS_Z && IDX) {
        if ((a.flags & DM_FLAG_A_SHARED) && sp) {
            hipLaunchKernelGGL((scan_bwd_chunked_kernel<T, TBC, true, true, 1, BWD_CHUNKED_NW, LC, true>), grid, block, 0, st, m);
            return;
        }
    }
    if constexpr (!HAS_Z && IDX) {                    // hoisted gate + hoisted softplus (the DiffMa mixer's call pattern)
        if (a.flags & DM_FLAG_DELTA_ACTIVATED) {
            hipLaunchKernelGGL((scan_bwd_chunked_kernel<T, TBC, false, true, 2, BWD_CHUNKED_NW, LC>), grid, block, 0, st, m);
            return;
        }
    }
    if (sp) hipLaunchKernelGGL((scan_bwd_chunked_kernel<T, TBC, HAS_Z, IDX, 1, BWD_CHUNKED_NW, LC>), grid, block, 0, st, m);
    else hipLaunchKernelGGL((scan_bwd_chunked_kernel<T, TBC, HAS_Z, IDX, 0, BWD_CHUNKED_NW, LC>), grid, block, 0, st, m);
}

template <typename T, typename TBC>
static int launch_bwd_chunked(const dm_scan_bwd_args& a, hipStream_t st, int lc) {
    const bool idx = a.z_row_index != nullptr;
#define DM_BWDC(HZ, IX)                                                                       \
    do {                                                                                      \
        if (lc == BWD_CHUNKED_LC_LONG) launch_bwd_chunked3<T, TBC, HZ, IX, BWD_CHUNKED_LC_LONG>(a, st); \
        else launch_bwd_chunked3<T, TBC, HZ, IX, BWD_CHUNKED_LC_SHORT>(a, st);                  \
    } while (0)
    if (a.z) {
        if (idx) DM_BWDC(true, true);
        else DM_BWDC(true, false);
    } else {
        if (idx) DM_BWDC(false, true);
        else DM_BWDC(false, false);
    }
#undef DM_BWDC
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_selective_scan_bwd (chunked): launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T>
static int bwd_dispatch(const dm_scan_bwd_args& a, hipStream_t st) {
    const int lc = bwd_chunked_lc(a.nseq, a.dim, a.seqlen, a.dstate, a.flags);
    if (lc > 0) {
        if (a.bc_dtype == DM_F32) return launch_bwd_chunked<T, float>(a, st, lc);
        if (a.bc_dtype == a.io_dtype) return launch_bwd_chunked<T, T>(a, st, lc);
    }
    return bwd_dispatch_bc<T>(a, st);
}

}  // namespace dm
