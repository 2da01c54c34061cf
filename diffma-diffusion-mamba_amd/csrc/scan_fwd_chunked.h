#pragma once
// K1c  chunk-parallel forward scan for SMALL launches (sampling and graphed small-batch training at batch 1..~10).
//
// The sequential kernel (scan_fwd_impl.h) needs nseq * dim/64 >= ~2000 waves to fill the chip; a p_sample step at
// batch 8 launches 384, and every launch then costs the full 196-step dependent chain (~70 us, 38 % of a
// graph-replayed step).  Here the time axis is cut into NW chunks, one WAVE per chunk inside a workgroup that owns
// (sequence, 64 channels):
//   pass 1  every wave runs the recurrence over its LC steps from h = 0 (no C, no z, no output) and publishes, per
//           state, its local end state and the chunk's total decay  P = exp(A * sum(delta))          -> LDS
//   combine after one barrier wave c folds the chunks before it:  H <- P_j * H + h_j ,  j = 0 .. c-1   (16 FMAs per chunk)
//   pass 2  every wave re-runs its chunk from the true entry state H and writes the gated outputs.
// The dependent chain is (0.75 + 1) * L/NW steps instead of L at 1.75x the arithmetic -- the trade the north star's
// "wavefront-parallel scan" asks for, taken only where latency (not throughput) is the limit.  Inputs of the chunk
// stay in registers between the passes; B/C rows of the chunk sit in a wave-private LDS slab.
#include <cstdlib>
#include "scan_fwd_impl.h"

namespace dm {

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX, bool SOFTPLUS, int NW, int LC, bool ASH = false, bool CKPT = false>
__global__ __launch_bounds__(64 * NW) void scan_fwd_chunked_kernel(const mix_args<dm_scan_fwd_args> pm) {
    const dm_scan_fwd_args& p = pm.a[blockIdx.z];      // grid.z = congruent launches sharing this one (the two mixers of a block)
    constexpr int NP = N / 2;
    constexpr int ES = (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) float bc_lds[NW][LC][2 * N];      // [B row | C row] of every step of the wave's chunk
    __shared__ float xch_lds[NW][2][N][WAVE];                                   // [chunk][P | h_end][state][lane]
    const int lane = threadIdx.x & 63;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);             // wave = chunk index (kept scalar)
    const int d_raw = blockIdx.x * WAVE + lane;
    const int d = (d_raw < p.dim) ? d_raw : p.dim - 1;                          // lanes past the end shadow the last channel
    const int s = blockIdx.y;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int grp = (blockIdx.x * WAVE) / (p.dim / p.ngroups);
    const int l0 = c * LC;
    const int nl = (L - l0 < LC) ? ((L - l0 > 0) ? L - l0 : 0) : LC;            // steps of this chunk (0 for waves past the end)

    const rsrc_t r_u = make_rsrc((const T*)p.u + (int64_t)s * p.u_ss);
    const rsrc_t r_dt = make_rsrc((const T*)p.delta + (int64_t)s * p.dt_ss);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_o = make_rsrc((T*)p.out + (int64_t)s * p.o_ss);
    const int vo = d * ES;
    const int sl_u = (int)p.u_sl * ES, sl_dt = (int)p.dt_sl * ES, sl_z = (int)p.z_sl * ES, sl_o = (int)p.o_sl * ES;
    const cptr<int32_t> zidx = IDX ? as_const(p.z_row_index + (int64_t)dir * L) : nullptr;
    const cptr<int32_t> oidx = IDX ? as_const(p.out_row_index + (int64_t)dir * L) : nullptr;
    const TBC* __restrict__ Bg = (const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg;
    const TBC* __restrict__ Cg = (const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg;
    constexpr bool CK_PACKED = std::is_same<T, bf16_t>::value;
    constexpr int CK_ROWS = CK_PACKED ? NP : N;
    const int nck = (L + FWD_CKE - 1) / FWD_CKE;
    const rsrc_t r_ck = make_rsrc(CKPT ? (const uint32_t*)p.ckpt + (int64_t)s * nck * CK_ROWS * p.dim : nullptr);

    // ---- this chunk's inputs: requested up front, kept in registers for both passes ------------------------------
    float uu[LC], dl[LC], zz[LC];
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        const int l = (l0 + j < L) ? l0 + j : L - 1;
        uu[j] = bio<T>::ld(r_u, vo, l * sl_u);
        dl[j] = bio<T>::ld(r_dt, vo, l * sl_dt);
        zz[j] = HAS_Z ? bio<T>::ld(r_z, vo, (IDX ? zidx[l] : l) * sl_z) : 0.0f;
    }
    // B/C rows of the chunk -> wave-private LDS slab (LC*2N values, cooperatively)
    for (int e = lane; e < LC * 2 * N; e += WAVE) {
        const int j = e / (2 * N), cc = e % (2 * N);
        const int l = (l0 + j < L) ? l0 + j : L - 1;
        bc_lds[c][j][cc] = (cc < N) ? io<TBC>::ld(Bg + (int64_t)l * p.B_sl + cc) : io<TBC>::ld(Cg + (int64_t)l * p.C_sl + cc - N);
    }
    f32x2 A2[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        A2[k].x = p.A[(int64_t)d * N + 2 * k] * LOG2E;
        A2[k].y = p.A[(int64_t)d * N + 2 * k + 1] * LOG2E;
    }
    const float Dv = p.D ? p.D[d] : 0.0f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.0f;
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        float x = dl[j] + bias;
        if (SOFTPLUS) x = softplus_f(x);
        dl[j] = (j < nl) ? x : 0.0f;                     // steps past the end: decay 1, input 0 => exact no-ops
        uu[j] = (j < nl) ? uu[j] : 0.0f;
    }

    // ---- pass 1: local end state and total decay ------------------------------------------------------------------
    f32x2 h[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) h[k] = (f32x2){0.0f, 0.0f};
    float sd = 0.0f;
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        const float du = dl[j] * uu[j];
        sd += dl[j];
        float a_sh = 0.0f;
        if (ASH) a_sh = fast_exp2(A2[0].x * dl[j]);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            f32x2 a, bb;
            if (ASH) {
                a = (f32x2){a_sh, a_sh};
            } else {
                const f32x2 t = A2[k] * dl[j];
                a.x = fast_exp2(t.x);
                a.y = fast_exp2(t.y);
            }
            bb.x = bc_lds[c][j][2 * k];
            bb.y = bc_lds[c][j][2 * k + 1];
            h[k] = a * h[k] + bb * du;
        }
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const f32x2 t = A2[k] * sd;
        xch_lds[c][0][2 * k][lane] = fast_exp2(t.x);
        xch_lds[c][0][2 * k + 1][lane] = fast_exp2(t.y);
        xch_lds[c][1][2 * k][lane] = h[k].x;
        xch_lds[c][1][2 * k + 1][lane] = h[k].y;
    }
    __syncthreads();

    // ---- combine: the state entering this chunk ---------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NP; ++k) h[k] = (f32x2){0.0f, 0.0f};
    for (int j = 0; j < c; ++j) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            f32x2 pj, hj;
            pj.x = xch_lds[j][0][2 * k][lane];
            pj.y = xch_lds[j][0][2 * k + 1][lane];
            hj.x = xch_lds[j][1][2 * k][lane];
            hj.y = xch_lds[j][1][2 * k + 1][lane];
            h[k] = pj * h[k] + hj;
        }
    }

    // ---- pass 2: the chunk again from its true entry state, with outputs ----------------------------------------------
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        if (j < nl) {                                    // wave-uniform
            const int l = l0 + j;
            float Bc[N], Cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                Bc[k] = bc_lds[c][j][k];
                Cc[k] = bc_lds[c][j][N + k];
            }
            const float y = scan_step<N, HAS_Z, false, ASH>(h, A2, Bc, Cc, uu[j], dl[j], zz[j], Dv, 0.0f);
            bio<T>::st(r_o, vo, (IDX ? oidx[l] : l) * sl_o, y);
            if (CKPT && ((l + 1) % FWD_CKE == 0 || l + 1 == L)) {       // training: the state entering every 4-step chunk (wave-uniform);
                const int ci = (l + 1 < L) ? (l + 1) / FWD_CKE : 0;      // slot 0 = the state after the last step
                if constexpr (CK_PACKED) {                               // [chunk][N/8][d][4 words], see scan_fwd_impl.h
                    uint32_t w[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[k]) : "v"(h[k].x), "v"(h[k].y));
                    bio_st_words<NP>(w, r_ck, d * 16, ci * p.dim * NP * 4, p.dim * 16);
                } else {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        bio<float>::st(r_ck, d * 4, ((ci * N + 2 * k) * p.dim) * 4, h[k].x);
                        bio<float>::st(r_ck, d * 4, ((ci * N + 2 * k + 1) * p.dim) * 4, h[k].y);
                    }
                }
            }
        }
    }
}

constexpr int CHUNKED_NW = 14, CHUNKED_LC = 14;          // 14 waves x 14 steps: L <= 196 (DiffMa's 14x14 token grid)

// true if the launch is small enough that latency, not throughput, is the limit (and the variant is instantiated).
// The 137 KB of LDS allow one workgroup per CU, i.e. 256 at a time: measured (bf16, dim 1024, L 196) 75 -> 24 us at
// nseq 3, 78 -> 50 us at nseq 24 (two rounds), break-even near nseq 48.
static inline bool use_chunked_fwd(const dm_scan_fwd_args& a) {
    static const int env = [] { const char* e = getenv("DM_SCAN_CHUNKED"); return e ? atoi(e) : -1; }();   // 0 / 1: developer override
    const int forced = (a.flags & DM_FLAG_SCAN_SEQUENTIAL) ? 0 : ((a.flags & DM_FLAG_SCAN_CHUNKED) ? 1 : env);
    if (forced == 0 || (a.flags & DM_FLAG_OUT_ACCUMULATE)) return false;
    const int64_t waves = (int64_t)a.nseq * ((a.dim + WAVE - 1) / WAVE);
    // checkpoints: built for the two model call patterns only (gated + softplus in the scan; gate and softplus hoisted out of it)
    if (a.ckpt && !(a.z_row_index && ((a.z && (a.flags & DM_FLAG_DELTA_SOFTPLUS)) || (!a.z && !(a.flags & DM_FLAG_DELTA_SOFTPLUS))))) return false;
    return !a.last_state && a.dstate == 16 && (waves <= 512 || forced == 1) && a.seqlen > 4 * CHUNKED_NW &&
           a.seqlen <= CHUNKED_NW * CHUNKED_LC;
}

template <typename T, typename TBC, bool HAS_Z, bool IDX>
static void launch_fwd_chunked2(const dm_scan_fwd_args& a, hipStream_t st) {
    unsigned gz;
    const mix_args<dm_scan_fwd_args> m = mix_make(a, gz);
    dim3 grid((a.dim + WAVE - 1) / WAVE, a.nseq, gz), block(WAVE * CHUNKED_NW);
    const bool sp = (a.flags & DM_FLAG_DELTA_SOFTPLUS) != 0;
    if constexpr (HAS_Z && IDX) {                   // the model's call pattern: also built with checkpoints (small-batch training)
        if ((a.flags & DM_FLAG_A_SHARED) && sp) {
            if (a.ckpt) hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, true, true, true, CHUNKED_NW, CHUNKED_LC, true, true>), grid, block, 0, st, m);
            else hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, true, true, true, CHUNKED_NW, CHUNKED_LC, true, false>), grid, block, 0, st, m);
            return;
        }
        if (a.ckpt && sp) {
            hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, true, true, true, CHUNKED_NW, CHUNKED_LC, false, true>), grid, block, 0, st, m);
            return;
        }
    }
    if constexpr (!HAS_Z && IDX) {                  // hoisted gate + hoisted softplus (DM_FLAG_DELTA_ACTIVATED arrives here as "no softplus, no bias")
        if (a.ckpt && !sp) {
            hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, false, true, false, CHUNKED_NW, CHUNKED_LC, false, true>), grid, block, 0, st, m);
            return;
        }
    }
    if (sp) hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, HAS_Z, IDX, true, CHUNKED_NW, CHUNKED_LC>), grid, block, 0, st, m);
    else hipLaunchKernelGGL((scan_fwd_chunked_kernel<T, TBC, 16, HAS_Z, IDX, false, CHUNKED_NW, CHUNKED_LC>), grid, block, 0, st, m);
}

template <typename T, typename TBC>
static int launch_fwd_chunked(const dm_scan_fwd_args& a, hipStream_t st) {
    const bool idx = a.z_row_index != nullptr;
    if (a.z) {
        if (idx) launch_fwd_chunked2<T, TBC, true, true>(a, st);
        else launch_fwd_chunked2<T, TBC, true, false>(a, st);
    } else {
        if (idx) launch_fwd_chunked2<T, TBC, false, true>(a, st);
        else launch_fwd_chunked2<T, TBC, false, false>(a, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_selective_scan_fwd (chunked): launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

template <typename T>
static int dispatch_fwd(const dm_scan_fwd_args& a, hipStream_t st) {
    if (use_chunked_fwd(a)) {
        if (a.bc_dtype == DM_F32) return launch_fwd_chunked<T, float>(a, st);
        if (a.bc_dtype == a.io_dtype) return launch_fwd_chunked<T, T>(a, st);
    }
    return dispatch_bc<T>(a, st);
}

}  // namespace dm
