#pragma once
// K2  dm_selective_scan_bwd -- Mamba-1 selective scan, backward (reverse time), gfx950.
//
// Replaces selective_scan_cuda.bwd (autograd of mamba_inner_fn / selective_scan_fn on the training
// path, reference train.py:259).  Equations: SURVEY.md A.1-bwd.
//
// Layout: token-major like the forward, but a lane owns a (channel, state-slice) pair: with SPLIT = 4 the
// 64 lanes of a wave are 16 channels x 4 slices of d_state/4 states.  The backward needs, per lane, the
// CK recomputed in-chunk states (hs[j] = state before step j), the adjoint carry, the dA accumulator and
// the chunk's inputs; a whole channel per lane (16 states) is ~270 live VGPRs and spills, a 4-state slice is
// ~100 and runs at 4+ waves/SIMD.  The price is a 2-step quad (DPP) sum of y, G.B and the dA-term per time
// step and a replicated softplus/silu per slice.
//
// Per chunk of CK = 8 steps (the forward saved the state entering every chunk):
//   1. reload the state slice, recompute the CK in-chunk states in registers,
//   2. walk the chunk backwards carrying  carry_n = a_{j+1,n} * dL/dh_{j+1,n},
//      accumulating dA / dD / dbias per lane (written once as per-sequence partials, no atomics),
//   3. dB/dC: the per-lane products are reduced over the wave's 16 channels with two permlane swaps and two
//      DPP rotations, staged in LDS, summed over the 4 waves of the workgroup after the chunk (one barrier
//      per chunk) and stored as one row of partials per (step, 64-channel workgroup); the dim/64 workgroups
//      of a sequence are summed by the caller (deterministic).
// One sweep over u, delta, z, dout (read) and du, ddelta, dz (write): 28 B/element in fp32 + checkpoints.
#include "dm_common.h"

namespace dm {

constexpr int BWD_CK = 8;      // must equal the forward's ckpt_every
constexpr int BWD_SUB = 8;     // steps whose recomputed states are held at once (CK = one level, CK/2 = two-level recompute)
constexpr int BWD_WAVES = 4;   // waves per workgroup

// ---- cross-lane helpers ----------------------------------------------------------------------
__device__ __forceinline__ void swap32(float& a, float& b) {   // a[lanes 32..63] <-> b[lanes 0..31]
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {   // odd 16-lane rows of a <-> even rows of b
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_SWAP = 0xB1;     // [1,0,3,2]  lane ^ 1
constexpr int DPP_QUAD_HALF = 0x4E;     // [2,3,0,1]  lane ^ 2
constexpr int DPP_ROW_ROR = 0x120;      // + n : rotate right by n inside a 16-lane row

// sum over the SPLIT consecutive lanes that share a channel (every lane gets the total)
template <int SPLIT>
__device__ __forceinline__ float slice_sum(float x) {
    if (SPLIT >= 2) x += dpp<DPP_QUAD_SWAP>(x);
    if (SPLIT >= 4) x += dpp<DPP_QUAD_HALF>(x);
    return x;
}

// v[0 .. M) per lane (M = 2*NS).  Sums over the CW = 64/SPLIT channels of the wave, i.e. over all lanes with
// the same slice index q.  After the call register i (< M/4) of a lane holds the total of value
//     4*i + 2*b4 + b5      (b5, b4 = bits 5, 4 of the lane id)
// replicated over the lanes of its 16-lane row that share q.
template <int M, int SPLIT>
__device__ __forceinline__ void channel_reduce(float (&v)[M]) {
    static_assert(M % 4 == 0, "need at least 2 states per lane");
#pragma unroll
    for (int i = 0; i < M / 2; ++i) { swap32(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
#pragma unroll
    for (int i = 0; i < M / 4; ++i) { swap16(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
#pragma unroll
    for (int i = 0; i < M / 4; ++i) {
        v[i] += dpp<DPP_ROW_ROR + 8>(v[i]);
        if (SPLIT <= 4) v[i] += dpp<DPP_ROW_ROR + 4>(v[i]);
        if (SPLIT <= 2) v[i] += dpp<DPP_ROW_ROR + 2>(v[i]);
        if (SPLIT <= 1) v[i] += dpp<DPP_ROW_ROR + 1>(v[i]);
    }
}

// make a value opaque to the optimiser (costs no instruction): stops it from keeping the exp() / B-row
// values of the recompute pass alive across the whole chunk just to save recomputing them
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ int opaque_i(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <typename T, typename TBC, int N, int SPLIT, bool HAS_Z, bool IDX, bool SOFTPLUS>
__global__ __launch_bounds__(64 * BWD_WAVES) void scan_bwd_kernel(const dm_scan_bwd_args p) {
    constexpr int NS = N / SPLIT, NPL = NS / 2, CW = WAVE / SPLIT, CK = BWD_CK, SUB = BWD_SUB, M = 2 * NS, R = M / 4;
    constexpr int ES = (int)sizeof(T);
    static_assert(N % SPLIT == 0 && NS % 2 == 0, "d_state/SPLIT must be even");
    static_assert(R <= 16 / SPLIT, "not enough stager lanes per row");
    static_assert(CK % SUB == 0, "chunk must be a whole number of sub-chunks");
    __shared__ float red_lds[2][BWD_WAVES][CK][2 * N];
    __shared__ __attribute__((aligned(16))) float bc_lds[2][CK][2 * N];   // [B row | C row] of every step of a chunk, all waves share a sequence

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int c = lane / SPLIT, q = lane % SPLIT;
    const int d_raw = (blockIdx.x * BWD_WAVES + wave) * CW + c;
    const bool active = d_raw < p.dim;
    const int d = active ? d_raw : p.dim - 1;
    const int s = blockIdx.y;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int grp = (blockIdx.x * BWD_WAVES * CW) / (p.dim / p.ngroups);
    const int nwg = gridDim.x;
    const int nchunk = (L + CK - 1) / CK;

    // one SRD per tensor, based at this sequence; row offsets are wave-uniform byte offsets in SGPRs
    const rsrc_t r_u = make_rsrc((const T*)p.u + (int64_t)s * p.u_ss);
    const rsrc_t r_dt = make_rsrc((const T*)p.delta + (int64_t)s * p.dt_ss);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_g = make_rsrc((const T*)p.dout + (int64_t)(IDX ? sb : s) * p.do_ss);
    const rsrc_t r_du = make_rsrc((T*)p.du + (int64_t)s * p.du_ss);
    const rsrc_t r_ddt = make_rsrc((T*)p.ddelta + (int64_t)s * p.ddt_ss);
    const rsrc_t r_dz = make_rsrc(HAS_Z ? (T*)p.dz + (int64_t)s * p.dz_ss : nullptr);
    const TBC* __restrict__ Bg = (const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg;
    const TBC* __restrict__ Cg = (const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg;
    const rsrc_t r_ck = make_rsrc(p.ckpt ? p.ckpt + (int64_t)s * nchunk * N * p.dim : nullptr);
    const int vo = d * ES;                 // per-lane byte offset of the channel, shared by all T tensors
    const int vo_ck = d * 4;
    const int sl_u = (int)p.u_sl * ES, sl_dt = (int)p.dt_sl * ES, sl_z = (int)p.z_sl * ES, sl_g = (int)p.do_sl * ES;
    const int sl_du = (int)p.du_sl * ES, sl_ddt = (int)p.ddt_sl * ES, sl_dz = (int)p.dz_sl * ES;
    const int i_B_sl = (int)p.B_sl, i_C_sl = (int)p.C_sl;
    const cptr<int32_t> zidx = IDX ? as_const(p.z_row_index + (int64_t)dir * L) : nullptr;
    const cptr<int32_t> oidx = IDX ? as_const(p.out_row_index + (int64_t)dir * L) : nullptr;

    f32x2 A2[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        A2[k].x = p.A[(int64_t)d * N + q * NS + 2 * k] * LOG2E;
        A2[k].y = p.A[(int64_t)d * N + q * NS + 2 * k + 1] * LOG2E;
    }
    const float Dv = p.D ? p.D[d] : 0.0f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.0f;

    f32x2 carry[NPL], dA[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) { carry[k] = (f32x2){0.f, 0.f}; dA[k] = (f32x2){0.f, 0.f}; }
    float dD_acc = 0.f, dbias_acc = 0.f;

    // which reduced register this lane stages in LDS, and where
    const int jrow = (lane & 15) / SPLIT;                         // index of the lane among its row's same-q lanes
    const int vidx = 4 * jrow + 2 * ((lane >> 4) & 1) + ((lane >> 5) & 1);      // value index in [0, M) if jrow < R
    const int col = (vidx < NS) ? (q * NS + vidx) : (N + q * NS + vidx - NS);   // [dB(0..N) | dC(0..N)]
    const bool stager = jrow < R;

    // B/C rows of a chunk: CK*2N values, fetched cooperatively (one or two per thread), one chunk ahead
    constexpr int BC_PER_THREAD = (CK * 2 * N + 64 * BWD_WAVES - 1) / (64 * BWD_WAVES);
    auto fetch_bc = [&](int chunk, float(&v)[BC_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < BC_PER_THREAD; ++i) {
            const int e = tid + i * 64 * BWD_WAVES;
            const int j = e / (2 * N), cc = e % (2 * N);
            int l = chunk * CK + j;
            l = (l < L) ? l : L - 1;
            v[i] = 0.f;
            if (e < CK * 2 * N) v[i] = (cc < N) ? io<TBC>::ld(Bg + l * i_B_sl + cc) : io<TBC>::ld(Cg + l * i_C_sl + cc - N);
        }
    };
    auto stash_bc = [&](int b, const float(&v)[BC_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < BC_PER_THREAD; ++i) {
            const int e = tid + i * 64 * BWD_WAVES;
            if (e < CK * 2 * N) bc_lds[b][e / (2 * N)][e % (2 * N)] = v[i];
        }
    };
    {
        float v[BC_PER_THREAD];
        fetch_bc(nchunk - 1, v);
        stash_bc(0, v);
    }
    __syncthreads();

    int buf = 0;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int l0 = ch * CK;
        float bc_next[BC_PER_THREAD];
        if (ch > 0) fetch_bc(ch - 1, bc_next);          // lands while this chunk computes
        // ---- chunk inputs (invalid tail steps become exact no-ops: dl = u = g = 0) -------------------
        float uu[CK], dl[CK], zz[CK], gg[CK];
        int zrow[CK];
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const int l = (l0 + j < L) ? l0 + j : L - 1;
            zrow[j] = IDX ? zidx[l] : l;
            const int orow = IDX ? oidx[l] : l;
            uu[j] = bio<T>::ld(r_u, vo, l * sl_u);
            dl[j] = bio<T>::ld(r_dt, vo, l * sl_dt);
            zz[j] = HAS_Z ? bio<T>::ld(r_z, vo, zrow[j] * sl_z) : 0.f;
            gg[j] = bio<T>::ld(r_g, vo, orow * sl_g);
        }
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const bool valid = (l0 + j) < L;
            float x = dl[j] + bias;
            if (SOFTPLUS) x = softplus_f(x);
            dl[j] = valid ? x : 0.f;
            uu[j] = valid ? uu[j] : 0.f;
            gg[j] = (valid && active) ? gg[j] : 0.f;
        }
        // ---- state slice entering the chunk ---------------------------------------------------------
        f32x2 h0[NPL];
        if (ch == 0) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) h0[k] = (f32x2){0.f, 0.f};
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                h0[k].x = bio<float>::ld(r_ck, vo_ck, ((ch * N + q * NS + 2 * k) * p.dim) * 4);
                h0[k].y = bio<float>::ld(r_ck, vo_ck, ((ch * N + q * NS + 2 * k + 1) * p.dim) * 4);
            }
        }

        // one forward step of the slice: h <- a*h + B*dl*u   (used by all three recompute passes)
        auto fwd_step = [&](f32x2(&h)[NPL], int j) {
            float Bv[NS];
            const float* brow = &bc_lds[0][0][0] + opaque_i((buf * CK + j) * 2 * N + q * NS);   // re-read, do not keep rows in VGPRs
#pragma unroll
            for (int k = 0; k < NS; ++k) Bv[k] = brow[k];
            const float dlo = opaque(dl[j]);
            const float du = dlo * uu[j];
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const f32x2 t = A2[k] * dlo;
                f32x2 a;
                a.x = fast_exp2(t.x);
                a.y = fast_exp2(t.y);
                f32x2 bb;
                bb.x = Bv[2 * k];
                bb.y = Bv[2 * k + 1];
                h[k] = a * h[k] + bb * du;
            }
        };

        // Two-level recompute: sub-chunks of SUB steps, last one first.  The states of ONE sub-chunk live
        // in registers (hs); earlier sub-chunks are re-advanced from the chunk's entry state when needed.
#pragma unroll
        for (int sc = CK / SUB - 1; sc >= 0; --sc) {
            f32x2 h[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) h[k] = h0[k];
#pragma unroll
            for (int j = 0; j < sc * SUB; ++j) fwd_step(h, j);           // advance to the sub-chunk start
            f32x2 hs[SUB][NPL];                                          // hs[i] = state before step sc*SUB+i
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
#pragma unroll
                for (int k = 0; k < NPL; ++k) hs[i][k] = h[k];
                fwd_step(h, sc * SUB + i);
            }
            // ---- reverse sweep over the sub-chunk (h = state AFTER step j at the top of iteration j) -------
#pragma unroll
            for (int i = SUB - 1; i >= 0; --i) {
                const int j = sc * SUB + i;
                const int lraw = l0 + j;
                const bool valid = lraw < L;                    // wave-uniform
                const int l = valid ? lraw : L - 1;
                float Bv[NS], Cv[NS];
                const float* brow = &bc_lds[0][0][0] + opaque_i((buf * CK + j) * 2 * N + q * NS);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    Bv[k] = brow[k];
                    Cv[k] = brow[N + k];
                }
                const float g = gg[j];
                float sz = 1.f, gy = g;
                if (HAS_Z) {
                    sz = sigmoid_f(zz[j]);
                    gy = g * zz[j] * sz;
                }
                const float dlo = opaque(dl[j]);
                const float du = dlo * uu[j];
                f32x2 yp2 = (f32x2){0.f, 0.f}, GB2 = (f32x2){0.f, 0.f}, dlA2 = (f32x2){0.f, 0.f};
                float red[M];
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    f32x2 bb, cc;
                    bb.x = Bv[2 * k]; bb.y = Bv[2 * k + 1];
                    cc.x = Cv[2 * k]; cc.y = Cv[2 * k + 1];
                    const f32x2 t = A2[k] * dlo;
                    f32x2 a;
                    a.x = fast_exp2(t.x);
                    a.y = fast_exp2(t.y);
                    const f32x2 hj = h[k];
                    const f32x2 hp = hs[i][k];
                    yp2 += cc * hj;
                    const f32x2 G = cc * gy + carry[k];          // dL/dh_j
                    const f32x2 dCp = hj * gy;
                    carry[k] = a * G;                            // a_j * dL/dh_j, flows to step j-1
                    const f32x2 Gt = carry[k] * hp;              // = G * a * h_{j-1}
                    dlA2 += A2[k] * Gt;
                    dA[k] += Gt * dlo;
                    GB2 += G * bb;
                    const f32x2 dBp = G * du;
                    red[2 * k] = dBp.x;
                    red[2 * k + 1] = dBp.y;
                    red[NS + 2 * k] = dCp.x;
                    red[NS + 2 * k + 1] = dCp.y;
                    h[k] = hp;
                }
                const float ypre = slice_sum<SPLIT>(yp2.x + yp2.y) + Dv * uu[j];
                const float GB = slice_sum<SPLIT>(GB2.x + GB2.y);
                const float dlA = slice_sum<SPLIT>(dlA2.x + dlA2.y);
                float ddl = uu[j] * GB + LN2 * dlA;
                const float duv = dlo * GB + gy * Dv;
                if (SOFTPLUS) ddl *= (1.0f - fast_exp2(-dlo * LOG2E));   // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
                if (q == 0) {                                             // one lane per channel owns the channel sums
                    dD_acc += gy * uu[j];
                    dbias_acc += ddl;
                }
                if (valid && active && q == 0) {
                    bio<T>::st(r_du, vo, l * sl_du, duv);
                    bio<T>::st(r_ddt, vo, l * sl_ddt, ddl);
                    if (HAS_Z) {
                        const float dzv = g * ypre * sz * (1.0f + zz[j] * (1.0f - sz));
                        bio<T>::st(r_dz, vo, zrow[j] * sl_dz, dzv);
                    }
                }
                channel_reduce<M, SPLIT>(red);
                float val = red[0];
#pragma unroll
                for (int r = 1; r < R; ++r) val = (jrow == r) ? red[r] : val;
                if (stager) red_lds[buf][wave][j][col] = val;
            }
        }
        // ---- sum the workgroup's waves' dB/dC rows of this chunk and store them --------------------------
        __syncthreads();
        for (int e = tid; e < CK * 2 * N; e += 64 * BWD_WAVES) {
            const int j = e / (2 * N), cc = e % (2 * N);
            if (l0 + j < L) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < BWD_WAVES; ++w) acc += red_lds[buf][w][j][cc];
                p.dBC_partial[(((int64_t)s * L + l0 + j) * nwg + blockIdx.x) * (2 * N) + cc] = acc;
            }
        }
        if (ch > 0) stash_bc(buf ^ 1, bc_next);
        __syncthreads();
        buf ^= 1;
    }
    if (active) {
        float* dAp = p.dA_partial + ((int64_t)s * p.dim + d) * N + q * NS;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            dAp[2 * k] = dA[k].x;
            dAp[2 * k + 1] = dA[k].y;
        }
        if (q == 0) {
            if (p.dD_partial) p.dD_partial[(int64_t)s * p.dim + d] = dD_acc;
            if (p.dbias_partial) p.dbias_partial[(int64_t)s * p.dim + d] = dbias_acc;
        }
    }
}

template <int N> struct bwd_split { static constexpr int value = (N >= 16) ? 2 : 1; };   // lanes per channel

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX>
static void launch_bwd2(const dm_scan_bwd_args& a, hipStream_t st, dim3 grid) {
    if (a.flags & DM_FLAG_DELTA_SOFTPLUS)
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, HAS_Z, IDX, true>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
    else
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, HAS_Z, IDX, false>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
}

template <typename T, typename TBC, int N>
static int launch_bwd(const dm_scan_bwd_args& a, hipStream_t st) {
    constexpr int WGCH = (WAVE / bwd_split<N>::value) * BWD_WAVES;   // channels per workgroup
    dim3 grid((a.dim + WGCH - 1) / WGCH, a.nseq);
    const bool idx = a.z_row_index != nullptr;
    if (a.z) {
        if (idx) launch_bwd2<T, TBC, N, true, true>(a, st, grid);
        else launch_bwd2<T, TBC, N, true, false>(a, st, grid);
    } else {
        if (idx) launch_bwd2<T, TBC, N, false, true>(a, st, grid);
        else launch_bwd2<T, TBC, N, false, false>(a, st, grid);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_selective_scan_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T, typename TBC>
static int bwd_dispatch_n(const dm_scan_bwd_args& a, hipStream_t st) {
    switch (a.dstate) {
        case 16: return launch_bwd<T, TBC, 16>(a, st);
#ifndef DM_FAST_BUILD
        case 8: return launch_bwd<T, TBC, 8>(a, st);
        case 32: return launch_bwd<T, TBC, 32>(a, st);
#endif
        default: set_error("dm_selective_scan_bwd: d_state=%d not instantiated (8,16,32)", a.dstate); return DM_ERR_DSTATE;
    }
}

template <typename T>
static int bwd_dispatch_bc(const dm_scan_bwd_args& a, hipStream_t st) {
    if (a.bc_dtype == DM_F32) return bwd_dispatch_n<T, float>(a, st);
    if (a.bc_dtype == a.io_dtype) return bwd_dispatch_n<T, T>(a, st);
    set_error("dm_selective_scan_bwd: bc_dtype %d must be fp32 or equal io_dtype %d", a.bc_dtype, a.io_dtype);
    return DM_ERR_DTYPE;
}

}  // namespace dm
