// K2  dm_selective_scan_bwd -- Mamba-1 selective scan, backward (reverse time), gfx950.
//
// Replaces selective_scan_cuda.bwd (autograd of mamba_inner_fn / selective_scan_fn on the training
// path, reference train.py:259).  Equations: SURVEY.md A.1-bwd.
//
// Same lane-per-channel, token-major design as the forward (scan_fwd.hip).  The forward saved the
// state entering every chunk of CK time steps; here each lane
//   1. reloads that state, recomputes the CK in-chunk states in registers (hs[j] = state before step j),
//   2. walks the chunk backwards carrying the adjoint  carry_n = a_{j+1,n} * dL/dh_{j+1,n},
//   3. accumulates dA, dD, dbias per lane (written once as per-sequence partials, no atomics),
//   4. reduces the per-lane dB/dC contributions of the wave's 64 channels with a permlane-swap / DPP
//      reduce-scatter (2N values -> one value per lane, ~70 VALU ops) and stores one row of partials per
//      step; the dim/64 waves of a sequence are summed by the caller (deterministic).
// Chunks are walked last-to-first, so the whole pass is one sweep over u, delta, z, dout (read) and
// du, ddelta, dz (write): 28 B/element in fp32, plus the checkpoints.
#include "dm_common.h"

namespace dm {

constexpr int BWD_CK = 8;   // must equal the forward's ckpt_every

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
// Involutions used inside a 16-lane row.  Each flips a prefix of the lane-bit set {3,2,1,0}, so the
// partner of a lane always sits on the same side of every EARLIER split -- which is all a
// reduce-scatter needs (the pairing need not be an xor).
constexpr int DPP_ROW_MIRROR = 0x140;       // l -> 15-l   (flips bits 3..0), split on bit 3
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // l -> 7-l    (flips bits 2..0), split on bit 2
constexpr int DPP_QUAD_REVERSE = 0x1B;      // [3,2,1,0]   (flips bits 1..0), split on bit 1
constexpr int DPP_QUAD_SWAP = 0xB1;         // [1,0,3,2]   (flips bit 0),     split on bit 0

template <int CTRL, int M>
__device__ __forceinline__ void scatter_level(float (&v)[M], int nregs, bool side) {
#pragma unroll
    for (int i = 0; i < M / 2; ++i) {
        if (i < nregs / 2) {
            const float p = v[2 * i] + dpp<CTRL>(v[2 * i]);
            const float q = v[2 * i + 1] + dpp<CTRL>(v[2 * i + 1]);
            v[i] = side ? q : p;
        }
    }
}

// Reduce-scatter M (= 2*d_state, 16/32/64) per-lane values over the 64 lanes of the wave.  On return
// v[0] of lane l holds the wave-wide sum of value  idx(l) = b5 + 2*b4 + 4*b3 + 8*b2 (+16*b1 (+32*b0)),
// b_i = bit i of l; lanes differing only in unused low bits hold the same sum.
template <int M>
__device__ __forceinline__ void wave_reduce_scatter(float (&v)[M], int lane) {
    static_assert(M == 16 || M == 32 || M == 64, "2*d_state must be 16, 32 or 64 in the backward kernel");
#pragma unroll
    for (int i = 0; i < M / 2; ++i) { swap32(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
#pragma unroll
    for (int i = 0; i < M / 4; ++i) { swap16(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
    scatter_level<DPP_ROW_MIRROR, M>(v, M / 4, (lane & 8) != 0);
    scatter_level<DPP_ROW_HALF_MIRROR, M>(v, M / 8, (lane & 4) != 0);
    if (M >= 32) scatter_level<DPP_QUAD_REVERSE, M>(v, M / 16, (lane & 2) != 0);
    else v[0] += dpp<DPP_QUAD_REVERSE>(v[0]);
    if (M >= 64) scatter_level<DPP_QUAD_SWAP, M>(v, M / 32, (lane & 1) != 0);
    else v[0] += dpp<DPP_QUAD_SWAP>(v[0]);
}
template <int M>
__device__ __forceinline__ int reduce_scatter_index(int lane) {
    int idx = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3);
    if (M >= 32) idx |= ((lane >> 1) & 1) << 4;
    if (M >= 64) idx |= (lane & 1) << 5;
    return idx;
}

template <typename TBC, int N>
__device__ __forceinline__ void load_row(float (&v)[N], const TBC* base, int sl, int l) {
    const cptr<TBC> r = as_const(base + l * sl);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = cio<TBC>::ld(r + k);
}

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX, bool SOFTPLUS>
__global__ __launch_bounds__(64, 2) void scan_bwd_kernel(const dm_scan_bwd_args p) {
    // The d_state states are processed in groups of GS = 8 (GP = 4 packed pairs): one group's CK
    // recomputed in-chunk states are CK*GS = 64 registers, which keeps the kernel at 2 waves/SIMD.
    constexpr int GS = 8, GP = GS / 2, NG = N / GS, NP = N / 2;
    static_assert(N % GS == 0, "d_state must be a multiple of 8");
    constexpr int CK = BWD_CK;
    const int lane = threadIdx.x;
    const int d0 = blockIdx.x * WAVE;
    const bool active = (d0 + lane) < p.dim;
    const int d = active ? d0 + lane : p.dim - 1;
    const int s = blockIdx.y;
    const int L = p.seqlen;
    // in-sequence row offsets fit 32 bits (validated by the host entry point)
    const int i_B_sl = (int)p.B_sl;
    const int i_C_sl = (int)p.C_sl;
    const int i_ddt_sl = (int)p.ddt_sl;
    const int i_do_sl = (int)p.do_sl;
    const int i_dt_sl = (int)p.dt_sl;
    const int i_du_sl = (int)p.du_sl;
    const int i_dz_sl = (int)p.dz_sl;
    const int i_u_sl = (int)p.u_sl;
    const int i_z_sl = (int)p.z_sl;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int grp = d0 / (p.dim / p.ngroups);
    const int nw = (p.dim + WAVE - 1) / WAVE;

    const T* __restrict__ up = (const T*)p.u + (int64_t)s * p.u_ss + d;
    const T* __restrict__ dp = (const T*)p.delta + (int64_t)s * p.dt_ss + d;
    const T* __restrict__ zp = HAS_Z ? (const T*)p.z + (int64_t)sb * p.z_ss + d : nullptr;
    const T* __restrict__ gp = (const T*)p.dout + (int64_t)(IDX ? sb : s) * p.do_ss + d;
    T* __restrict__ dup = (T*)p.du + (int64_t)s * p.du_ss + d;
    T* __restrict__ ddp = (T*)p.ddelta + (int64_t)s * p.ddt_ss + d;
    T* __restrict__ dzp = HAS_Z ? (T*)p.dz + (int64_t)s * p.dz_ss + d : nullptr;
    const TBC* Bp = (const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg;
    const TBC* Cp = (const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg;
    const cptr<int32_t> zidx = IDX ? as_const(p.z_row_index + (int64_t)dir * L) : nullptr;
    const cptr<int32_t> oidx = IDX ? as_const(p.out_row_index + (int64_t)dir * L) : nullptr;

    f32x2 A2[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        A2[k].x = p.A[(int64_t)d * N + 2 * k] * LOG2E;
        A2[k].y = p.A[(int64_t)d * N + 2 * k + 1] * LOG2E;
    }
    const float Dv = p.D ? p.D[d] : 0.0f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.0f;

    f32x2 carry[NP], dA[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { carry[k] = (f32x2){0.f, 0.f}; dA[k] = (f32x2){0.f, 0.f}; }
    float dD_acc = 0.f, dbias_acc = 0.f;

    const int nchunk = (L + CK - 1) / CK;
    const int widx = reduce_scatter_index<2 * GS>(lane);          // 0..15: [0,8) = dB, [8,16) = dC of the group
    const bool writer = (lane & 3) == 0;

    for (int c = nchunk - 1; c >= 0; --c) {
        const int l0 = c * CK;
        // ---- chunk inputs (invalid tail steps become exact no-ops: dl = u = g = 0) -------------------
        float uu[CK], dl[CK], zz[CK], gg[CK];
        {
            T ru[CK], rd[CK], rz[CK], rg[CK];
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                const int l = (l0 + j < L) ? l0 + j : L - 1;
                ru[j] = up[l * i_u_sl];
                rd[j] = dp[l * i_dt_sl];
                if (HAS_Z) rz[j] = zp[(IDX ? zidx[l] : l) * i_z_sl];
                rg[j] = gp[(IDX ? oidx[l] : l) * i_do_sl];
            }
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                const bool valid = (l0 + j) < L;
                uu[j] = valid ? io<T>::ld(&ru[j]) : 0.f;
                float x = io<T>::ld(&rd[j]) + bias;
                if (SOFTPLUS) x = softplus_f(x);
                dl[j] = valid ? x : 0.f;
                zz[j] = HAS_Z ? io<T>::ld(&rz[j]) : 0.f;
                gg[j] = (valid && active) ? io<T>::ld(&rg[j]) : 0.f;
            }
        }
        float gy[CK], ypre[CK], GBs[CK], dlAs[CK];
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            gy[j] = HAS_Z ? gg[j] * silu_f(zz[j]) : gg[j];
            ypre[j] = 0.f; GBs[j] = 0.f; dlAs[j] = 0.f;
        }

#pragma unroll
        for (int sg = 0; sg < NG; ++sg) {
            // ---- state of this group entering the chunk ------------------------------------------------
            f32x2 h[GP];
            if (c == 0) {
#pragma unroll
                for (int k = 0; k < GP; ++k) h[k] = (f32x2){0.f, 0.f};
            } else {
                const float* ck = p.ckpt + (((int64_t)s * nchunk + c) * N + sg * GS) * p.dim + d;
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    h[k].x = ck[(int64_t)(2 * k) * p.dim];
                    h[k].y = ck[(int64_t)(2 * k + 1) * p.dim];
                }
            }
            // ---- forward recompute: hs[j] = state before step j ------------------------------------------
            f32x2 hs[CK][GP];
#pragma unroll
            for (int j = 0; j < CK; ++j) {
                int l = (l0 + j < L) ? l0 + j : L - 1;
                asm volatile("" : "+s"(l));              // pins this step's scalar loads below the previous step
                float Bv[GS];
                load_row<TBC, GS>(Bv, Bp + sg * GS, i_B_sl, l);
                const float du = dl[j] * uu[j];
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    hs[j][k] = h[k];
                    const f32x2 t = A2[sg * GP + k] * dl[j];
                    f32x2 a;
                    a.x = fast_exp2(t.x);
                    a.y = fast_exp2(t.y);
                    f32x2 bb;
                    bb.x = Bv[2 * k];
                    bb.y = Bv[2 * k + 1];
                    h[k] = a * h[k] + bb * du;
                }
                __builtin_amdgcn_sched_barrier(0);   // keep one step's operands live at a time
            }
            // ---- reverse sweep (h = state AFTER step j at the top of iteration j) --------------------------
#pragma unroll
            for (int j = CK - 1; j >= 0; --j) {
                const int lraw = l0 + j;
                const bool valid = lraw < L;                    // wave-uniform
                int l = valid ? lraw : L - 1;
                asm volatile("" : "+s"(l));
                float Bv[GS], Cv[GS];
                load_row<TBC, GS>(Bv, Bp + sg * GS, i_B_sl, l);
                load_row<TBC, GS>(Cv, Cp + sg * GS, i_C_sl, l);
                const float du = dl[j] * uu[j];
                f32x2 yp2 = (f32x2){0.f, 0.f}, GB2 = (f32x2){0.f, 0.f}, dlA2 = (f32x2){0.f, 0.f};
                float red[2 * GS];
#pragma unroll
                for (int k = 0; k < GP; ++k) {
                    f32x2 bb, cc;
                    bb.x = Bv[2 * k]; bb.y = Bv[2 * k + 1];
                    cc.x = Cv[2 * k]; cc.y = Cv[2 * k + 1];
                    const f32x2 A2k = A2[sg * GP + k];
                    const f32x2 t = A2k * dl[j];
                    f32x2 a;
                    a.x = fast_exp2(t.x);
                    a.y = fast_exp2(t.y);
                    const f32x2 hj = h[k];
                    const f32x2 hp = hs[j][k];
                    yp2 += cc * hj;
                    const f32x2 G = cc * gy[j] + carry[sg * GP + k];   // dL/dh_j
                    const f32x2 dCp = hj * gy[j];
                    const f32x2 Gt = G * (a * hp);
                    dlA2 += A2k * Gt;
                    dA[sg * GP + k] += Gt * dl[j];
                    GB2 += G * bb;
                    const f32x2 dBp = G * du;
                    carry[sg * GP + k] = a * G;
                    red[2 * k] = dBp.x;
                    red[2 * k + 1] = dBp.y;
                    red[GS + 2 * k] = dCp.x;
                    red[GS + 2 * k + 1] = dCp.y;
                    h[k] = hp;
                }
                ypre[j] += yp2.x + yp2.y;
                GBs[j] += GB2.x + GB2.y;
                dlAs[j] += dlA2.x + dlA2.y;
                wave_reduce_scatter<2 * GS>(red, lane);
                if (valid && writer) {
                    const int col = (widx < GS) ? (sg * GS + widx) : (N + sg * GS + widx - GS);
                    p.dBC_partial[(((int64_t)s * L + l) * nw + blockIdx.x) * (2 * N) + col] = red[0];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- per-step input gradients --------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < CK; ++j) {
            const int lraw = l0 + j;
            const bool valid = lraw < L;
            const int l = valid ? lraw : L - 1;
            float ddl = uu[j] * GBs[j] + LN2 * dlAs[j];
            const float duv = dl[j] * GBs[j] + gy[j] * Dv;
            dD_acc += gy[j] * uu[j];
            if (SOFTPLUS) ddl *= (1.0f - fast_exp2(-dl[j] * LOG2E));   // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
            dbias_acc += ddl;
            if (valid && active) {
                io<T>::st(dup + l * i_du_sl, duv);
                io<T>::st(ddp + l * i_ddt_sl, ddl);
                if (HAS_Z) {
                    const float sz = sigmoid_f(zz[j]);
                    const float yfull = ypre[j] + Dv * uu[j];
                    const float dzv = gg[j] * yfull * sz * (1.0f + zz[j] * (1.0f - sz));
                    io<T>::st(dzp + (IDX ? zidx[l] : l) * i_dz_sl, dzv);
                }
            }
        }
    }
    if (active) {
        float* dAp = p.dA_partial + ((int64_t)s * p.dim + d) * N;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            dAp[2 * k] = dA[k].x;
            dAp[2 * k + 1] = dA[k].y;
        }
        if (p.dD_partial) p.dD_partial[(int64_t)s * p.dim + d] = dD_acc;
        if (p.dbias_partial) p.dbias_partial[(int64_t)s * p.dim + d] = dbias_acc;
    }
}

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX>
static void launch_bwd2(const dm_scan_bwd_args& a, hipStream_t st, dim3 grid) {
    if (a.flags & DM_FLAG_DELTA_SOFTPLUS)
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, HAS_Z, IDX, true>), grid, dim3(WAVE), 0, st, a);
    else
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, HAS_Z, IDX, false>), grid, dim3(WAVE), 0, st, a);
}

template <typename T, typename TBC, int N>
static int launch_bwd(const dm_scan_bwd_args& a, hipStream_t st) {
    dim3 grid((a.dim + WAVE - 1) / WAVE, a.nseq);
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

extern "C" int dm_selective_scan_bwd(const dm_scan_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_selective_scan_bwd: null args"); return DM_ERR_ARG; }
    const dm_scan_bwd_args& a = *args;
    if (!a.u || !a.delta || !a.dout || !a.A || !a.B || !a.C || !a.du || !a.ddelta || !a.dBC_partial || !a.dA_partial) {
        set_error("dm_selective_scan_bwd: null tensor pointer"); return DM_ERR_ARG;
    }
    if ((a.z != nullptr) != (a.dz != nullptr)) { set_error("dm_selective_scan_bwd: dz must be given iff z is"); return DM_ERR_ARG; }
    if (a.nseq <= 0 || a.dim <= 0 || a.seqlen <= 0 || a.ngroups <= 0) { set_error("dm_selective_scan_bwd: non-positive size"); return DM_ERR_ARG; }
    if (a.nseq > 65535) { set_error("dm_selective_scan_bwd: nseq %d > 65535", a.nseq); return DM_ERR_ARG; }
    if (a.ckpt_every != BWD_CK) { set_error("dm_selective_scan_bwd: ckpt_every must be %d", BWD_CK); return DM_ERR_ARG; }
    if (!a.ckpt && a.seqlen > BWD_CK) { set_error("dm_selective_scan_bwd: ckpt required for seqlen > %d", BWD_CK); return DM_ERR_ARG; }
    if (a.u_sd != 1 || a.dt_sd != 1 || a.do_sd != 1 || a.du_sd != 1 || a.ddt_sd != 1 || (a.z && (a.z_sd != 1 || a.dz_sd != 1)) ||
        a.B_sn != 1 || a.C_sn != 1) {
        set_error("dm_selective_scan_bwd: needs token-major tensors (channel stride 1, state stride 1)"); return DM_ERR_LAYOUT;
    }
    if (a.dim % a.ngroups != 0 || (a.ngroups > 1 && (a.dim / a.ngroups) % WAVE != 0)) {
        set_error("dm_selective_scan_bwd: dim/ngroups must be a multiple of 64"); return DM_ERR_LAYOUT;
    }
    if (a.batch_per_dir > 0 && a.nseq % a.batch_per_dir != 0) { set_error("dm_selective_scan_bwd: nseq %% batch_per_dir != 0"); return DM_ERR_ARG; }
    if ((a.z_row_index == nullptr) != (a.out_row_index == nullptr)) {
        set_error("dm_selective_scan_bwd: z_row_index and out_row_index must both be set or both be NULL"); return DM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (a.io_dtype) {
        case DM_F32: return bwd_dispatch_bc<float>(a, st);
        case DM_BF16: return bwd_dispatch_bc<bf16_t>(a, st);
        case DM_F16: return bwd_dispatch_bc<f16_t>(a, st);
        default: set_error("dm_selective_scan_bwd: bad io_dtype %d", a.io_dtype); return DM_ERR_DTYPE;
    }
}
