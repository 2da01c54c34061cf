#pragma once
// K1  dm_selective_scan_fwd -- Mamba-1 selective scan, forward, for gfx950 (MI355X).
//
// Replaces selective_scan_cuda.fwd behind selective_scan_fn / mamba_inner_fn
// (reference call sites block/mamba.py:11, 346-348; mathematics SURVEY.md A.1 step 4).
//
// Design (CDNA4-first, not the upstream CUDA tiling):
//   * token-major tensors [seq][l][d]: ONE LANE PER CHANNEL, one wave64 per 64 channels of one
//     sequence.  Every per-step access of the wave is a single coalesced 256-B row segment, so there
//     is no LDS transposition and no cross-lane traffic at all.
//   * the recurrence runs sequentially in time inside the lane with the d_state states held in
//     registers (packed f32x2 -> v_pk_mul/v_pk_fma): per (b,d,l) element that is N exp2 + ~2.5N packed
//     VALU ops, i.e. the work-optimal count -- a wave-parallel associative (Blelloch) scan of the same
//     recurrence costs ~2.5x the VALU work and is ALU-bound below the HBM roof.  Measured (round 2 PMC):
//     71 VALU instructions per wave-step in the fp32 loop against 67 the mathematics needs, pipe 81 % busy at 1.69 GHz
//     (round 3: the 15 further EXECUTED instructions per step were the waterfall loop of the B/C fetch; removed).
//   * B_l / C_l are shared by all channels of a sequence: the rows of a block of 8 steps are fetched with ONE vector load
//     per lane, parked as fp32 in a wave-private LDS slab and read back per step as broadcast ds_read_b128 (the scalar-cache
//     route of the first version cost 32 SALU unpack operations per step for 16-bit B/C).
//   * latency hiding comes from a register prefetch ring of PF time steps (u, delta, z rows are
//     requested PF steps before use), not from occupancy: at batch 64 there is one wave per SIMD.
//   * CrossScan's z gather and CrossMerge's inverse reindex are folded into the row addressing
//     (z_row_index / out_row_index), so the (B,3,2D,L) buffer of block/mamba.py:41 never exists.
#include "dm_common.h"
#include <type_traits>

namespace dm {

constexpr int FWD_CKE = 4;     // steps between checkpoints (the backward's sub-chunk length)
#ifndef DM_FWD_F32_WAVES
#define DM_FWD_F32_WAVES 1     // minimum waves per SIMD requested for the fp32-I/O instantiations (register budget 512 / waves)
#endif

// One time step of the recurrence for one lane.
// ASH (DM_FLAG_A_SHARED): every state of the channel decays with the same factor -> one exp per step.
template <int N, bool HAS_Z, bool SOFTPLUS, bool ASH = false>
__device__ __forceinline__ float scan_step(f32x2 (&h)[N / 2], const f32x2 (&A2)[N / 2], const float (&Bv)[N],
                                           const float (&Cv)[N], float uu, float draw, float zz, float Dv, float bias) {
    float dl = draw + bias;
    if (SOFTPLUS) dl = softplus_f(dl);
    const float du = dl * uu;
    f32x2 acc = (f32x2){0.0f, 0.0f};
    float a_sh = 0.0f;
    if (ASH) a_sh = fast_exp2(A2[0].x * dl);
#pragma unroll
    for (int k = 0; k < N / 2; ++k) {
        f32x2 a;
        if (ASH) {
            a = (f32x2){a_sh, a_sh};
        } else {
            const f32x2 t = A2[k] * dl;
            a.x = fast_exp2(t.x);
            a.y = fast_exp2(t.y);
        }
        f32x2 bb, cc;
        bb.x = Bv[2 * k];
        bb.y = Bv[2 * k + 1];
        cc.x = Cv[2 * k];
        cc.y = Cv[2 * k + 1];
        h[k] = a * h[k] + bb * du;
        acc += h[k] * cc;
    }
    float y = acc.x + acc.y + Dv * uu;
    if (HAS_Z) y *= silu_f(zz);
    return y;
}

// IDX : z_row_index / out_row_index tables are used (both non-null)
// CKPT: the state is written to p.ckpt after every FWD_CKE = 4 steps (fp32, or bf16 pairs for bf16 I/O)
// ACC : DM_FLAG_OUT_ACCUMULATE -- the launch ADDS its outputs to what `out` already holds (read-add-store through
//       out_row_index) instead of storing them.  The host walks the directions of the CrossMerge with one launch each into ONE
//       token-order buffer (direction 0 stores, the others accumulate): the merge sum happens here and the separate
//       4-tensor merge pass disappears.  One direction per launch, so no two waves ever touch the same output element.
template <typename T, typename TBC, int N, bool HAS_Z, bool IDX, bool CKPT, bool SOFTPLUS, int PF, bool ASH = false, bool ACC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 2 && sizeof(TBC) == 2 && N <= 16) ? (ACC ? 3 : 4) : ((sizeof(T) == 4 && N <= 16) ? DM_FWD_F32_WAVES : 1)))) void scan_fwd_kernel(const dm_scan_fwd_args p) {
    static_assert(!ACC || IDX, "accumulating launches are built for the model's call pattern (row-index tables)");
    static_assert(N % 2 == 0, "d_state must be even");
    static_assert(PF == 8, "the B/C staging below maps 8 steps onto the 64 lanes");
    constexpr int NP = N / 2;
    constexpr int ES = (int)sizeof(T), EBC = (int)sizeof(TBC);
    // B_l / C_l rows are shared by every channel of a sequence.  The wave fetches the rows of a whole
    // block of PF steps with ONE vector load per lane (lane = step*8 + part; N/4 consecutive values each),
    // parks them in its private 2 x (PF*2N*4 B) LDS slab as fp32, and every step reads its [B | C] row back
    // with broadcast ds_read_b128.  Nothing here is wave-shared, so no barrier is needed.
    constexpr int PER = N / 4;                       // B/C values fetched per lane per block
    __shared__ __attribute__((aligned(16))) float bc_lds[2][PF][2 * N];
    const int d_raw = blockIdx.x * WAVE + threadIdx.x;
    // Lanes past the last channel shadow channel dim-1: they help fetching B/C and then compute and store
    // exactly the same values to exactly the same addresses as that lane (a benign duplicate store), which
    // keeps every store unpredicated.
    const int d = (d_raw < p.dim) ? d_raw : p.dim - 1;
    const int L = p.seqlen;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int grp = (blockIdx.x * WAVE) / (p.dim / p.ngroups);
    const int s = blockIdx.y;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;

    // SRD addressing: one descriptor per tensor based at this sequence, the lane's channel offset in ONE
    // VGPR, wave-uniform row offsets in SGPRs (dm_common.h).
    const rsrc_t r_u = make_rsrc((const T*)p.u + (int64_t)s * p.u_ss);
    const rsrc_t r_dt = make_rsrc((const T*)p.delta + (int64_t)s * p.dt_ss);
    const rsrc_t r_z = make_rsrc(HAS_Z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_o = make_rsrc((T*)p.out + (int64_t)s * p.o_ss);
    const int vo = d * ES;
    const int sl_u = (int)p.u_sl * ES, sl_dt = (int)p.dt_sl * ES, sl_z = (int)p.z_sl * ES, sl_o = (int)p.o_sl * ES;
    const int lane = threadIdx.x;
    const int bc_step = lane >> 3, bc_part = lane & 7;                    // part 0..3 -> B, 4..7 -> C
    const bool bc_isB = bc_part < 4;
    // The B-lanes and the C-lanes of the wave read different tensors (two descriptors: a 2-trip waterfall loop per block of 8
    // steps) and every lane group a different ROW: the row offset therefore rides in the per-lane VGPR offset.  (Rounds 1-2 passed
    // it as the scalar offset, which made the waterfall loop run once per (tensor, row) = 16 trips of 8 VALU instructions per
    // block -- the 15 instructions per step that round 2's accounting found executed but not in the static loop count, 86.4 vs 71.)
    const rsrc_t r_bc = make_rsrc(bc_isB ? (const void*)((const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg)
                                         : (const void*)((const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg));
    const int sl_bc = (int)(bc_isB ? p.B_sl : p.C_sl) * EBC;
    const int vo_bc = (bc_part & 3) * PER * EBC;
    float* const bc_slot = &bc_lds[0][bc_step][(bc_isB ? 0 : N) + (bc_part & 3) * PER];   // + buf*PF*2N
    const cptr<int32_t> zidx = IDX ? as_const(p.z_row_index + (int64_t)dir * L) : nullptr;
    const cptr<int32_t> oidx = IDX ? as_const(p.out_row_index + (int64_t)dir * L) : nullptr;

    // per-channel constants: A pre-scaled by log2(e) so that exp(delta*A) = v_exp_f32(delta*A2)
    f32x2 A2[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        A2[j].x = p.A[(int64_t)d * N + 2 * j] * LOG2E;
        A2[j].y = p.A[(int64_t)d * N + 2 * j + 1] * LOG2E;
    }
    const float Dv = p.D ? p.D[d] : 0.0f;
    const float bias = p.delta_bias ? p.delta_bias[d] : 0.0f;

    f32x2 h[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) h[j] = (f32x2){0.0f, 0.0f};

    constexpr bool CK_PACKED = std::is_same<T, bf16_t>::value;      // checkpoint = pairs of bf16 in one 32-bit word
    constexpr int CK_ROWS = CK_PACKED ? NP : N;                     // 32-bit rows of [dim] per checkpoint
    const int nchunk = CKPT ? (L + FWD_CKE - 1) / FWD_CKE : 0;
    const rsrc_t r_ck = make_rsrc(CKPT ? (const uint32_t*)p.ckpt + (int64_t)s * nchunk * CK_ROWS * p.dim : nullptr);
    auto store_slot = [&](int c) {
        if constexpr (CK_PACKED) {          // [chunk][N/8][d][4 words]: two dense 16-byte stores per lane (8 dword stores until round 3)
            uint32_t w[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[k]) : "v"(h[k].x), "v"(h[k].y));
            bio_st_words<NP>(w, r_ck, d * 16, c * p.dim * NP * 4, p.dim * 16);
        } else {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                bio<float>::st(r_ck, d * 4, ((c * N + 2 * k) * p.dim) * 4, h[k].x);
                bio<float>::st(r_ck, d * 4, ((c * N + 2 * k + 1) * p.dim) * 4, h[k].y);
            }
        }
    };
    auto store_ckpt = [&](int done) {                               // done = steps finished (wave-uniform)
        if (done < L) store_slot(done / FWD_CKE);                   // state entering chunk done / FWD_CKE
    };

    // ---- register prefetch ring: rows of block b+1 are requested before block b is computed ----
    // (ACC: the value already in the output row rides in the same ring)
    auto ld_old = [&](int l) -> float {
        if constexpr (ACC) return bio<T>::ld(r_o, vo, oidx[l] * sl_o);
        else return 0.0f;
    };
    float ru[PF], rd[PF], rz[PF], ro[ACC ? PF : 1];
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int l = (j < L) ? j : L - 1;
        ru[j] = bio<T>::ld(r_u, vo, l * sl_u);
        rd[j] = bio<T>::ld(r_dt, vo, l * sl_dt);
        if (HAS_Z) rz[j] = bio<T>::ld(r_z, vo, (IDX ? zidx[l] : l) * sl_z);
        if (ACC) ro[j] = ld_old(l);
    }
    auto fetch_bc = [&](int l0, float(&v)[PER]) {        // rows l0 .. l0+PF-1 (clamped), this lane's piece
        int l = l0 + bc_step;
        l = (l < L) ? l : L - 1;
        bio_ld_vec<TBC, PER>(v, r_bc, vo_bc + l * sl_bc, 0);
    };
    auto stash_bc = [&](int b, const float(&v)[PER]) {
#pragma unroll
        for (int k = 0; k < PER; ++k) bc_slot[b * PF * 2 * N + k] = v[k];
    };
    {
        float v[PER];
        fetch_bc(0, v);
        stash_bc(0, v);
    }
    int buf = 0;

    // One block of PF steps: request the NEXT block's rows into (nu, nd, nz, nbc), then run the PF steps on
    // the rows that are already here.  The caller alternates two register sets (ping-pong), so the ring
    // never needs a register-to-register copy.
    auto run_block = [&](int l0, const float(&cu)[PF], const float(&cd)[PF], const float(&cz)[PF], const float(&co)[ACC ? PF : 1],
                         float(&nu)[PF], float(&nd)[PF], float(&nz)[PF], float(&no)[ACC ? PF : 1]) {
        float nbc[PER];
        fetch_bc(l0 + PF, nbc);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            int l = l0 + PF + j;
            l = (l < L) ? l : L - 1;
            nu[j] = bio<T>::ld(r_u, vo, l * sl_u);
            nd[j] = bio<T>::ld(r_dt, vo, l * sl_dt);
            if (HAS_Z) nz[j] = bio<T>::ld(r_z, vo, (IDX ? zidx[l] : l) * sl_z);
            if (ACC) no[j] = ld_old(l);
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int l = l0 + j;
            float Bc[N], Cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                Bc[k] = bc_lds[buf][j][k];
                Cc[k] = bc_lds[buf][j][N + k];
            }
            float y = scan_step<N, HAS_Z, SOFTPLUS, ASH>(h, A2, Bc, Cc, cu[j], cd[j], HAS_Z ? cz[j] : 0.0f, Dv, bias);
            if (ACC) y += co[j];
            bio<T>::st(r_o, vo, (IDX ? oidx[l] : l) * sl_o, y);
            if (CKPT && (j + 1) % FWD_CKE == 0) store_ckpt(l0 + j + 1);      // l0 is a multiple of PF
        }
        buf ^= 1;
        stash_bc(buf, nbc);
    };

    const int Lfull = (L / PF) * PF;
    float su[PF], sd[PF], sz[PF], so[ACC ? PF : 1];             // second register set of the ring
    int l0 = 0;
    for (; l0 + 2 * PF <= Lfull; l0 += 2 * PF) {
        run_block(l0, ru, rd, rz, ro, su, sd, sz, so);
        run_block(l0 + PF, su, sd, sz, so, ru, rd, rz, ro);
    }
    if (l0 < Lfull) {                          // odd number of full blocks
        run_block(l0, ru, rd, rz, ro, su, sd, sz, so);
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            ru[j] = su[j];
            rd[j] = sd[j];
            if (HAS_Z) rz[j] = sz[j];
            if (ACC) ro[j] = so[j];
        }
    }
    // tail (< PF steps); its rows are already in the ring
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const int l = Lfull + j;
        if (l < L) {
            float Bc[N], Cc[N];
#pragma unroll
            for (int k = 0; k < N; ++k) {
                Bc[k] = bc_lds[buf][j][k];
                Cc[k] = bc_lds[buf][j][N + k];
            }
            float y = scan_step<N, HAS_Z, SOFTPLUS, ASH>(h, A2, Bc, Cc, ru[j], rd[j], HAS_Z ? rz[j] : 0.0f, Dv, bias);
            if (ACC) y += ro[j];
            bio<T>::st(r_o, vo, (IDX ? oidx[l] : l) * sl_o, y);
            if (CKPT && (j + 1) % FWD_CKE == 0) store_ckpt(l + 1);
        }
    }

    if (CKPT) store_slot(0);                   // slot 0 (chunk 0 starts from zero) holds the state after the last step

    if (p.last_state) {
        float* ls = p.last_state + ((int64_t)s * N) * p.dim + d;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            ls[(int64_t)(2 * k) * p.dim] = h[k].x;
            ls[(int64_t)(2 * k + 1) * p.dim] = h[k].y;
        }
    }
}

constexpr int SCAN_PF = 8;
static_assert(SCAN_PF % FWD_CKE == 0, "checkpoints fall on fixed positions of a prefetch block");

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX>
static void launch_fwd3(const dm_scan_fwd_args& a, hipStream_t st, dim3 grid) {
    const bool sp = (a.flags & DM_FLAG_DELTA_SOFTPLUS) != 0;
    if constexpr (N == 16 && HAS_Z && IDX) {          // accumulating launch: the model's call pattern only
        if ((a.flags & DM_FLAG_OUT_ACCUMULATE) && sp && !(a.flags & DM_FLAG_A_SHARED)) {
            if (a.ckpt) hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, true, true, true, true, SCAN_PF, false, true>), grid, dim3(WAVE), 0, st, a);
            else hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, true, true, false, true, SCAN_PF, false, true>), grid, dim3(WAVE), 0, st, a);
            return;
        }
    }
    if constexpr (N == 16 && HAS_Z && IDX) {          // the one-exp variant is built for the Mamba-2 call pattern only
        if ((a.flags & DM_FLAG_A_SHARED) && sp) {
            if (a.ckpt) hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, true, true, true, true, SCAN_PF, true>), grid, dim3(WAVE), 0, st, a);
            else hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, true, true, false, true, SCAN_PF, true>), grid, dim3(WAVE), 0, st, a);
            return;
        }
    }
    if (a.ckpt) {
        if (sp) hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, HAS_Z, IDX, true, true, SCAN_PF>), grid, dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, HAS_Z, IDX, true, false, SCAN_PF>), grid, dim3(WAVE), 0, st, a);
    } else {
        if (sp) hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, HAS_Z, IDX, false, true, SCAN_PF>), grid, dim3(WAVE), 0, st, a);
        else hipLaunchKernelGGL((scan_fwd_kernel<T, TBC, N, HAS_Z, IDX, false, false, SCAN_PF>), grid, dim3(WAVE), 0, st, a);
    }
}

template <typename T, typename TBC, int N>
static int launch_fwd(const dm_scan_fwd_args& a, hipStream_t st) {
    dim3 grid((a.dim + WAVE - 1) / WAVE, a.nseq);
    const bool idx = a.z_row_index != nullptr;   // validated: both tables or neither
    if (a.z) {
        if (idx) launch_fwd3<T, TBC, N, true, true>(a, st, grid);
        else launch_fwd3<T, TBC, N, true, false>(a, st, grid);
    } else {
        if (idx) launch_fwd3<T, TBC, N, false, true>(a, st, grid);
        else launch_fwd3<T, TBC, N, false, false>(a, st, grid);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("dm_selective_scan_fwd: launch failed: %s", hipGetErrorString(e));
        return DM_ERR_LAUNCH;
    }
    return DM_OK;
}

template <typename T, typename TBC>
static int dispatch_n(const dm_scan_fwd_args& a, hipStream_t st) {
    switch (a.dstate) {
        case 16: return launch_fwd<T, TBC, 16>(a, st);
#ifndef DM_FAST_BUILD
        case 8: return launch_fwd<T, TBC, 8>(a, st);
        case 32: return launch_fwd<T, TBC, 32>(a, st);
        case 64: return launch_fwd<T, TBC, 64>(a, st);
#endif
        default:
            set_error("dm_selective_scan_fwd: d_state=%d not instantiated (8,16,32,64)", a.dstate);
            return DM_ERR_DSTATE;
    }
}

template <typename T>
static int dispatch_bc(const dm_scan_fwd_args& a, hipStream_t st) {
    // B/C either share the I/O dtype or are fp32 (keeps the instantiation count bounded)
    if (a.bc_dtype == DM_F32) return dispatch_n<T, float>(a, st);
    if (a.bc_dtype == a.io_dtype) return dispatch_n<T, T>(a, st);
    set_error("dm_selective_scan_fwd: bc_dtype %d must be fp32 or equal io_dtype %d", a.bc_dtype, a.io_dtype);
    return DM_ERR_DTYPE;
}

}  // namespace dm
