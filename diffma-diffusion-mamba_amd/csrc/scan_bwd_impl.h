#pragma once
// K2  dm_selective_scan_bwd -- Mamba-1 selective scan, backward (reverse time), gfx950.
//
// Replaces selective_scan_cuda.bwd (autograd of mamba_inner_fn / selective_scan_fn on the training
// path, reference train.py:259).  Equations: SURVEY.md A.1-bwd.
//
// Layout: token-major like the forward.  A lane owns SPLIT-th of a channel's states: for d_state 16 a whole channel
// (SPLIT = 1, 64 channels per wave, 232 VGPRs, 2 waves/SIMD), for d_state 32 half of one (SPLIT = 2).  With the
// forward's checkpoints every 4 steps only 4 recomputed states per lane are live (64 VGPRs for 16 states); the earlier
// 8-step scheme needed ~270 VGPRs for a whole channel and therefore ran 2 lanes per channel, which replicated every
// per-channel instruction (softplus, silu', conversions, stores: a third of the issue slots) -- a whole channel per
// lane measures 11 % faster despite the lower occupancy.
//
// Per staging chunk of CK = 8 steps:
//   1. per SUB = 4 step sub-chunk: reload the state slice from the forward's checkpoint (every 4 steps; bf16 pairs
//      for bf16 I/O, i.e. the same bytes as fp32 every 8) and recompute the 4 in-chunk states into registers
//      instead of re-advancing from an earlier state,
//   2. walk the sub-chunk backwards carrying  carry_n = a_{j+1,n} * dL/dh_{j+1,n},
//      accumulating dA / dD / dbias per lane (written once as per-sequence partials, no atomics),
//   3. dB/dC: the per-lane products are reduce-scattered over the wave's 4 lane groups (16-bit I/O: two
//      v_mfma_f32_16x16x32_bf16 with 0/1 selector fragments on the otherwise idle matrix pipe; fp32 I/O: permlane
//      swaps), written to LDS per row position, and after the chunk (one barrier) summed over the row positions
//      of a slice and the 4 waves and stored as one row of partials per (step, workgroup = 256 channels at d_state 16);
//      the workgroups of a sequence are summed by the caller (deterministic).
// One sweep over u, delta, z, dout (read) and du, ddelta, dz (write): 28 B/element in fp32 + checkpoints.
//
// Memory pipeline (round 3): all global loads run one stage ahead of their use -- the inputs and the checkpoint of the NEXT
// sub-chunk, the B/C rows of the next chunk, the row-table entries two sub-chunks ahead -- and every memory instruction of the
// loop is issued unconditionally: lanes / steps / slots that must not take part use an out-of-range buffer offset (BIO_OOB in a
// 2 GB window: loads return 0, stores are dropped).  Both halves matter.  hipcc's s_waitcnt pass counts outstanding operations
// exactly only along straight-line code; at every join whose arms issued different numbers of memory operations (a uniform
// `if` around the stores of a tail step, a zero-or-load checkpoint, a conditional fetch) it falls back to the smaller count, and
// with ~50 operations in flight a prefetched value was then waited for with vmcnt(19) right after 20 newer loads had been
// issued -- i.e. for a load that had just left.  (profiles/r03_k2_experiments.txt #9.)
#include <type_traits>
#include "dm_common.h"

#ifndef DM_K2_HALVES
#define DM_K2_HALVES 1         // developer A/B: 0 = the whole channel's 16 states in one pass (rounds 3-5)
#endif
#ifndef DM_K2_DEFER_WRITE
#define DM_K2_DEFER_WRITE 0    // developer A/B: 0 = the lane-group totals are written to LDS right behind their MFMAs
#endif
#ifndef DM_K2_LDS_AHEAD
#define DM_K2_LDS_AHEAD 1      // developer A/B: 0 = a step's B / C rows are read at its top, 1 = read ahead in the sweep only, 2 = in the recompute too
#endif
#ifndef DM_K2_EXP
#define DM_K2_EXP 0            // developer timing experiments (bit mask; results are WRONG when non-zero): 1 no dB/dC reduction,
#endif                         // 2 no barriers / flush, 4 no checkpoint loads, 8 no du / ddelta stores, 16 no LDS B/C re-reads,
                               // 32 dB/dC products and conversions kept but no MFMA / LDS write / flush, 64 no u / delta / dy loads,
                               // 128 the two MFMAs of a state group replaced by four XORs (LDS write and flush kept): the MFMAs' own price

namespace dm {

constexpr int BWD_CK = 8;      // steps per staging chunk (B/C rows and dB/dC partials go through LDS once per chunk)
constexpr int BWD_SUB = 4;     // checkpoint spacing = steps whose recomputed states are held in registers at once
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

// v[0 .. M) per lane (M = 2*NS).  Reduce-scatter over the 4 lane groups (16-lane rows) of the wave: after the call
// register i (< M/4) of a lane holds the lane-group total of value
//     4*i + 2*b4 + b5      (b5, b4 = bits 5, 4 of the lane id)
// for its row position (lane & 15).  The sum over the row positions that share a slice is done later, in LDS.
template <int M>
__device__ __forceinline__ void lane_group_reduce(float (&v)[M]) {
    static_assert(M % 4 == 0, "need at least 2 states per lane");
#pragma unroll
    for (int i = 0; i < M / 2; ++i) { swap32(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
#pragma unroll
    for (int i = 0; i < M / 4; ++i) { swap16(v[2 * i], v[2 * i + 1]); v[i] = v[2 * i] + v[2 * i + 1]; }
}

// ---- matrix-pipe variant of the first two stages (16-bit I/O only) --------------------------------------
// v_mfma_f32_16x16x32_bf16 with a 0/1 selector as the A fragment sums the B fragment over the 4 lane groups:
// lane l supplies B[k = (l>>4, e)][j = l&15] = value e of lane l, and A[i = l&15][k = (l>>4, e)] = (e == i) gives
// D[i][j] = sum_g value_i(lane j + 16 g).  Two instructions (values 0..7, 8..15) leave register r of lane l
// holding the lane-group total of value 4*(l>>4) + r for row position l&15 -- the same reduce-scatter as the
// 12 permlane swaps + 12 adds of lane_group_reduce, on the otherwise idle matrix pipe.  The products are rounded to bf16
// first (they are re-rounded to the 16-bit I/O dtype later anyway); fp32 I/O keeps the exact VALU path.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    return dm_cvt_pk_bf16(lo, hi);
}
__device__ __forceinline__ void mfma_selectors(int lane, u32x4_t& a_lo, u32x4_t& a_hi) {
    const int i = lane & 15;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        a_lo[p] = ((i == 2 * p) ? 0x3F80u : 0u) | ((i == 2 * p + 1) ? 0x3F800000u : 0u);
        a_hi[p] = ((i == 2 * p + 8) ? 0x3F80u : 0u) | ((i == 2 * p + 9) ? 0x3F800000u : 0u);
    }
}
__device__ __forceinline__ f32x4 mfma_group_sum16(const u32x4_t& a_lo, const u32x4_t& a_hi, const u32x4_t& v_lo, const u32x4_t& v_hi) {
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_lo), __builtin_bit_cast(bf16x8_t, v_lo), d, 0, 0, 0);
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a_hi), __builtin_bit_cast(bf16x8_t, v_hi), d, 0, 0, 0);
    return d;
}
// make a value opaque to the optimiser (costs no instruction): stops it from keeping the exp() / B-row
// values of the recompute pass alive across the whole chunk just to save recomputing them
__device__ __forceinline__ float opaque(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ uint32_t opaque_u(uint32_t x) {
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ int opaque_i(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// LDS rows of the staged B / C values are RE-READ at every use instead of being kept in VGPRs across the chunk: the chunk's base
// address (an LDS pointer in one VGPR) is made opaque to the optimiser once per pass over a sub-chunk (the recompute and the sweep must not share row values: 24
// VGPRs held across both), and the row's place inside the chunk -- a
// compile-time constant in the unrolled loop -- rides in the ds_read's immediate offset: no address instruction per row.  (Rounds 3-5
// made the element OFFSET opaque instead: an s_or, a v_mov and a v_lshl_add per row read, 84 instructions per 8-step chunk.)
typedef __attribute__((address_space(3))) const float* lds_cfptr;
typedef __attribute__((address_space(3))) const f32x4* lds_cf4ptr;
__device__ __forceinline__ lds_cfptr lds_opaque(lds_cfptr chunk_base) {      // one register copy; taken once per pass (recompute / sweep) of a state group
    asm volatile("" : "+v"(chunk_base));
    return chunk_base;
}
template <int NS>
__device__ __forceinline__ void lds_ld_vec(float (&v)[NS], lds_cfptr row) {
    if constexpr (NS % 4 == 0) {
#pragma unroll
        for (int k = 0; k < NS / 4; ++k) {
            const f32x4 t = ((lds_cf4ptr)row)[k];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = row[k];
    }
}

// DMODE (delta mode): 0 = delta + bias used as is, 1 = softplus(delta + bias) (DM_FLAG_DELTA_SOFTPLUS), 2 = delta already holds
// softplus(raw + bias) (DM_FLAG_DELTA_ACTIVATED: the producer of delta applied it once per element instead of every scan
// direction twice); the returned ddelta is the gradient of the RAW value in every mode: softplus'(x) = 1 - exp(-softplus(x)).
template <typename T, typename TBC, int N, int SPLIT, bool HAS_Z, bool IDX, int DMODE, bool ASH = false>
__global__ __launch_bounds__(64 * BWD_WAVES) __attribute__((amdgpu_waves_per_eu((N / SPLIT <= 8 && N == 16) ? 3 : (N <= 16 ? 2 : 1)))) void scan_bwd_kernel(const dm_scan_bwd_args p) {
    constexpr int NS = N / SPLIT, NPL = NS / 2, CW = WAVE / SPLIT, CK = BWD_CK, SUB = BWD_SUB, M = 2 * NS, R = M / 4;
    constexpr int ES = (int)sizeof(T);
    constexpr bool MFMA_RED = std::is_same<T, bf16_t>::value && M % 16 == 0;   // dB/dC lane-group sums on the matrix pipe
    // HALVES: the lane's 16 states as two groups of 8 walked one after the other, the recomputed steps' decay factors reused by the sweep
    constexpr bool HALVES = (DM_K2_HALVES != 0) && MFMA_RED && SPLIT == 1 && N == 16;
    constexpr int NH = HALVES ? 2 : 1, NPH = NPL / NH;
    constexpr bool CACHE_A = HALVES && !ASH;
    // (the variants with z carry zz[] / sz / ypre as well: the 16 read-ahead registers spill there -- they keep the read at the step's top)
    constexpr bool LDS_AHEAD = HALVES && !HAS_Z && (DM_K2_LDS_AHEAD != 0), LDS_AHEAD_REC = LDS_AHEAD && (DM_K2_LDS_AHEAD >= 2);
    constexpr bool DEFER_WRITE = LDS_AHEAD && (DM_K2_DEFER_WRITE != 0);
    static_assert(N % SPLIT == 0 && NS % 2 == 0, "d_state/SPLIT must be even");
    static_assert(CK % SUB == 0, "chunk must be a whole number of sub-chunks");
    // lane-group totals of the dB/dC products, one R-float slot per lane; every 16-lane row is shifted by 2R floats so
    // that the strided reads of the chunk-end summation spread over all LDS banks
    // (registers go to LDS in groups of 4: [group][lane][4] keeps every ds_write_b128 and the strided flush reads conflict-free)
    static_assert(R % 4 == 0, "lane-group totals are stored 4 registers at a time");
    constexpr int RED_HALF = WAVE * 4 + 4 * 8;
    constexpr int RED_ROW = (R / 4) * RED_HALF;
    __shared__ __attribute__((aligned(16))) float red_lds[BWD_WAVES][CK][RED_ROW];
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
    const rsrc_t r_g = make_rsrc((const T*)p.dout + (int64_t)((IDX && !(p.flags & DM_FLAG_DOUT_PER_SEQ)) ? sb : s) * p.do_ss);
    // (stores, checkpoint loads and the dB/dC partial rows go through 2 GB windows: predication by an out-of-range offset, no branch)
    const rsrc_t r_du = make_rsrc_2g((T*)p.du + (int64_t)s * p.du_ss);
    const rsrc_t r_ddt = make_rsrc_2g((T*)p.ddelta + (int64_t)s * p.ddt_ss);
    const rsrc_t r_dz = make_rsrc_2g(HAS_Z ? (T*)p.dz + (int64_t)s * p.dz_ss : nullptr);
    const TBC* __restrict__ Bg = (const TBC*)p.B + (int64_t)s * p.B_ss + (int64_t)grp * p.B_sg;
    const TBC* __restrict__ Cg = (const TBC*)p.C + (int64_t)s * p.C_ss + (int64_t)grp * p.C_sg;
    // checkpoints: the state entering every SUB-step sub-chunk; fp32 rows [n][d] or, for bf16 I/O, rows [n/2][d] of bf16 pairs
    constexpr bool CK_PACKED = std::is_same<T, bf16_t>::value;
    constexpr int CK_ROWS = CK_PACKED ? N / 2 : N;
    const int nck = (L + SUB - 1) / SUB;
    const rsrc_t r_ck = make_rsrc_2g(p.ckpt ? (const uint32_t*)p.ckpt + (int64_t)s * nck * CK_ROWS * p.dim : nullptr);
    const rsrc_t r_dbc = make_rsrc_2g(p.dBC_partial + (int64_t)s * L * gridDim.x * (2 * N));
    const int vo = d * ES;                 // per-lane byte offset of the channel, shared by all T tensors
    const int vo_st = (active && q == 0) ? vo : BIO_OOB;       // the lane that owns the channel's du / ddelta / dz
    // (fp32 rows [n][d]; packed: [N/8][d][4 words], 16 bytes per lane and access.  The slice offset is per lane, so it lives in the VGPR part of the address)
    const int vo_ck = CK_PACKED ? (d * 4 + q * (CK_ROWS / SPLIT / 4) * p.dim * 4) * 4 : (d + q * (CK_ROWS / SPLIT) * p.dim) * 4;
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

    u32x4_t sel_lo, sel_hi;
    if (MFMA_RED) mfma_selectors(lane, sel_lo, sel_hi);
    const int red_slot = lane * 4 + (lane >> 4) * 8;                 // + RED_HALF per group of 4 registers

    // B/C rows of a chunk: CK*2N values, fetched cooperatively (one or two per thread), one chunk ahead
    constexpr int BC_PER_THREAD = (CK * 2 * N + 64 * BWD_WAVES - 1) / (64 * BWD_WAVES);
    // (raw words, one unconditional load per element through a selected pointer: the conversion happens in stash_bc, a chunk later.
    //  Until round 3 the load sat in a conditional block together with its bf16 -> fp32 shift, the compiler kept the two together
    //  and every chunk began with `global_load_ushort; s_waitcnt vmcnt(0)` -- a full memory latency, with everything older drained.)
    auto fetch_bc = [&](int chunk, uint32_t(&v)[BC_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < BC_PER_THREAD; ++i) {
            int e = tid + i * 64 * BWD_WAVES;
            e = (e < CK * 2 * N) ? e : CK * 2 * N - 1;
            const int j = e / (2 * N), cc = e % (2 * N);
            int l = chunk * CK + j;
            l = (l < L) ? l : L - 1;
            const TBC* src = (cc < N) ? Bg + l * i_B_sl + cc : Cg + l * i_C_sl + (cc - N);
            v[i] = io<TBC>::ld_raw(src);
        }
    };
    auto stash_bc = [&](int b, const uint32_t(&v)[BC_PER_THREAD]) {
#pragma unroll
        for (int i = 0; i < BC_PER_THREAD; ++i) {
            const int e = tid + i * 64 * BWD_WAVES;
            if (e < CK * 2 * N) bc_lds[b][e / (2 * N)][e % (2 * N)] = io<TBC>::cv(v[i]);
        }
    };
    // sum chunk `chunk`'s staged products over the workgroup's waves and over the row positions of a slice,
    // and store its dB/dC partial rows
    auto flush_dbc = [&](int chunk) {
        const int l0 = chunk * CK;
        if constexpr (MFMA_RED && SPLIT == 1 && N == 16) {
            // 16-byte reads: a thread owns a QUAD of outputs (4 consecutive registers of a lane slot are 4 consecutive values) and a
            // quarter of the 16 row positions; the four quarter-owners are neighbouring lanes and meet through two quad-permute adds.
            // 16 ds_read_b128 per thread and chunk instead of 64 ds_read_b32 (8 -> 2 LDS instructions per step; the dword form also ran
            // a 2-way bank conflict between the wave's two steps).
            static_assert(CK * 2 * N == 64 * BWD_WAVES, "one output per thread");
            const int tq = tid & 3, g4 = (tid >> 2) & 7, j = tid >> 5;
            const float* base = &red_lds[0][j][(g4 >> 2) * RED_HALF + (16 * (g4 & 3) + 4 * tq) * 4 + (g4 & 3) * 8];
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < BWD_WAVES; ++w)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc4 += *reinterpret_cast<const f32x4*>(base + w * (CK * RED_ROW) + t * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc4[i];
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
                v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
                acc4[i] = v;
            }
            const float mine = tq == 0 ? acc4[0] : (tq == 1 ? acc4[1] : (tq == 2 ? acc4[2] : acc4[3]));             // value 4 g4 + tq of step j
            // whole-channel form: value V = column V of [dB 0..15 | dC 0..15].  HALVES: LDS group hf = V >> 4 holds [dB 8hf..8hf+7 | dC 8hf..8hf+7]
            const int V = 4 * g4 + tq;
            const int col = HALVES ? ((V & 8) ? N + 8 * (V >> 4) + (V & 7) : 8 * (V >> 4) + (V & 7)) : V;
            const int vo_p = (l0 + j < L) ? (((l0 + j) * nwg + (int)blockIdx.x) * (2 * N) + col) * 4 : BIO_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mine), r_dbc, vo_p, 0, 0);
            return;
        }
#pragma unroll
        for (int it = 0; it < BC_PER_THREAD; ++it) {
            const int e_ = tid + it * 64 * BWD_WAVES;
            const bool in = e_ < CK * 2 * N;
            const int e = in ? e_ : 0;
            const int j = e / (2 * N), cc = e % (2 * N);
            // rows past the end of the sequence (and threads past the chunk's values) store out of range: dropped, no branch
            const int vo_p = (in && l0 + j < L) ? (((l0 + j) * nwg + (int)blockIdx.x) * (2 * N) + cc) * 4 : BIO_OOB;
            float acc = 0.f;
            const int n = (cc < N) ? cc : cc - N;
            const int qq = n / NS;
            const int vidx = (cc < N) ? n % NS : NS + n % NS;                 // value index in [0, M) inside slice qq
            const int lane0 = (MFMA_RED ? 16 * ((vidx & 15) >> 2) : 32 * (vidx & 1) + 16 * ((vidx >> 1) & 1)) + qq;
            const int reg = MFMA_RED ? 4 * (vidx >> 4) + (vidx & 3) : (vidx >> 2);
#pragma unroll
            for (int w = 0; w < BWD_WAVES; ++w)
#pragma unroll
                for (int t = 0; t < 16 / SPLIT; ++t) acc += red_lds[w][j][(reg >> 2) * RED_HALF + (lane0 + SPLIT * t) * 4 + (lane0 >> 4) * 8 + (reg & 3)];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc), r_dbc, vo_p, 0, 0);
        }
    };
    {
        uint32_t v[BC_PER_THREAD];
        fetch_bc(nchunk - 1, v);
        stash_bc(0, v);
    }
    __syncthreads();

    // state entering sub-chunk ci, as raw checkpoint words (packed checkpoints stay raw until they are used: unpacking
    // at the load would put a vmcnt(0) right behind it).  Slot ci of the forward's checkpoints for 0 < 4 ci < L, slot 0
    // (= the state after the last step) for 4 ci == L, zero for ci == 0 and past the end.
    constexpr int H0W = CK_PACKED ? NPL : 2 * NPL;               // 32-bit words per state slice
    auto load_state = [&](int ci, uint32_t(&w)[H0W]) {
        const int slot_ = (ci > 0 && ci * SUB < L) ? ci : ((ci > 0 && ci * SUB == L) ? 0 : -1);      // wave-uniform
        const int vo_ = (slot_ < 0 || !p.ckpt) ? BIO_OOB : vo_ck;                                    // no slot: the loads return 0 (no branch)
        const int slot = slot_ < 0 ? 0 : slot_;
        if constexpr (CK_PACKED) {
            bio_ld_words<H0W>(w, r_ck, vo_, slot * p.dim * (N / 2) * 4, p.dim * 16);       // dense 16-byte loads
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                w[2 * k] = __builtin_amdgcn_raw_buffer_load_b32(r_ck, vo_, ((slot * N + 2 * k) * p.dim) * 4, 0);
                w[2 * k + 1] = __builtin_amdgcn_raw_buffer_load_b32(r_ck, vo_, ((slot * N + 2 * k + 1) * p.dim) * 4, 0);
            }
        }
    };
    auto unpack_state = [&](f32x2(&h)[NPH], const uint32_t(&w)[H0W], int hf) {       // group hf of the slice: pairs hf*NPH ..
#pragma unroll
        for (int k = 0; k < NPH; ++k) {
            const int kk = hf * NPH + k;
            if constexpr (CK_PACKED) {
                h[k].x = __uint_as_float(w[kk] << 16);
                h[k].y = __uint_as_float(w[kk] & 0xffff0000u);
            } else {
                h[k].x = __uint_as_float(w[2 * kk]);
                h[k].y = __uint_as_float(w[2 * kk + 1]);
            }
        }
    };
    // ---- the software pipeline (round 3) ---------------------------------------------------------------------------------
    // Sub-chunks (SUB steps, one checkpoint) are processed last to first.  While sub-chunk ci is being recomputed and swept, the
    // u / delta / dout (/ z) values and the checkpoint of sub-chunk ci-1 are already in flight into their own registers -- RAW
    // words, converted only when ci-1 starts, so that no wait lands behind the loads -- and the row-table entries of sub-chunk
    // ci-2 into SGPRs.  Until round 3 a chunk's 24 input loads were issued at its top and consumed at once: every wave spent one
    // full memory latency per 8 steps in s_waitcnt (SQ_WAIT_ANY = 24 % of the wave cycles, the VALU 71 % busy at two waves
    // per SIMD; 15 % and 78 % with the pipeline).  Registers: 12 converted + 12 raw inputs (24 converted before), 3 checkpoint slices (end state | current | next;
    // 5 before with the chunk-level checkpoint prefetch).
    static_assert(SUB == 4, "row-table entries of a sub-chunk travel as one 4-dword scalar load");
    constexpr int NSC = CK / SUB;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    // Four single scalar loads per table and sub-chunk, addresses clamped to the last row, no branch.  (Until round 6 a 4-dword load
    // for whole sub-chunks and the four single loads for the ragged tail were BOTH issued -- hipcc if-converts the choice -- into the
    // same SGPR quad: the write-after-write hazard put an `s_waitcnt lgkmcnt(0)` right behind the first load, a full scalar-memory
    // latency per table and sub-chunk with the whole wave parked; writing the choice as a real branch parks it at the join instead.)
    auto load_rows = [&](cptr<int32_t> tab, int ci) -> i32x4_t {          // rows of the steps of sub-chunk ci (clamped to L - 1)
        const int lb = ci < 0 ? 0 : ci * SUB;
        i32x4_t r = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < SUB; ++i) {
            const int l = (lb + i < L) ? lb + i : L - 1;
            r[i] = IDX ? tab[(uint32_t)l] : l;                     // (unsigned: no sign extension of the index on the scalar unit)
        }
        return r;
    };
    typename bio<T>::raw_t ru[SUB], rd[SUB], rg[SUB], rz[HAS_Z ? SUB : 1];      // raw input words of the sub-chunk in flight
    uint32_t ck_nx[H0W];                                          // its checkpoint, raw
    i32x4_t orow_is = load_rows(oidx, nchunk * NSC - 1), zrow_is = {0, 0, 0, 0};       // row-table entries of the sub-chunk to be ISSUED next
    if (HAS_Z) zrow_is = load_rows(zidx, nchunk * NSC - 1);
    auto issue_sub = [&](int ci) {          // requests sub-chunk ci (rows in orow_is / zrow_is), then the row entries of ci - 1
        const int lb = ci < 0 ? 0 : ci * SUB;              // (ci = -1 after the first sub-chunk of the sequence: rows 0.., never used)
#pragma unroll
        for (int i = 0; i < SUB; ++i) {
            const int l = (lb + i < L) ? lb + i : L - 1;
            if (DM_K2_EXP & 64) { ru[i] = (typename bio<T>::raw_t)opaque_u(0x3f00u); rd[i] = (typename bio<T>::raw_t)opaque_u(0x3c23u); rg[i] = (typename bio<T>::raw_t)opaque_u(0x3dccu); if (HAS_Z) rz[i] = 0; continue; }
            ru[i] = bio<T>::ld_raw(r_u, vo, l * sl_u);
            rd[i] = bio<T>::ld_raw(r_dt, vo, l * sl_dt);
            if (HAS_Z) rz[i] = bio<T>::ld_raw(r_z, vo, zrow_is[i] * sl_z);
            rg[i] = bio<T>::ld_raw(r_g, vo, orow_is[i] * sl_g);
        }
        load_state((DM_K2_EXP & 4) ? -1 : ci, ck_nx);
        orow_is = load_rows(oidx, ci - 1);
        if (HAS_Z) zrow_is = load_rows(zidx, ci - 1);
    };
    // state entering the sub-chunk processed BEFORE this one = this one's end state.  HALVES keeps it UNPACKED (the sub-chunk's own
    // entering state hs[0] is handed on as it is: 16 shift / mask instructions per sub-chunk less for 8 more registers); the
    // whole-channel forms keep the raw words
    uint32_t ck_end[H0W];
    load_state(nchunk * NSC, ck_end);
    f32x2 hend[HALVES ? NH : 1][NPH];
    if constexpr (HALVES) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) unpack_state(hend[hf], ck_end, hf);
    }
    i32x4_t zrow_nx = zrow_is;                                   // z rows of the sub-chunk in flight (its dz stores need them)
    issue_sub(nchunk * NSC - 1);

    int buf = 0;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
        const int l0 = ch * CK;
        f32x4 pend = {0.f, 0.f, 0.f, 0.f};                 // DM_K2_DEFER_WRITE: lane-group totals on their way to LDS
        int pend_off = 0;
        bool pend_have = false;
        // this chunk's staged rows (the lane's state slice of every row): one LDS address for the whole chunk
        const lds_cfptr bc_chunk = (lds_cfptr)(&bc_lds[0][0][0]) + (buf * CK * 2 * N + q * NS);
        uint32_t bc_next[BC_PER_THREAD];
        fetch_bc(ch > 0 ? ch - 1 : 0, bc_next);         // lands while this chunk computes (unconditional: a branch would pin the wait to the load)
        // Sub-chunks of SUB steps, last one first; each starts from its own checkpoint and its SUB states live in
        // registers (hs) during its reverse sweep.
#pragma unroll
        for (int sc = NSC - 1; sc >= 0; --sc) {
            const int ci = ch * NSC + sc;
            // ---- take over the sub-chunk that was in flight (invalid tail steps become exact no-ops: g = 0 => every adjoint term is 0)
            float uu[SUB], dl[SUB], zz[SUB], gg[SUB];
            uint32_t ck_cur[H0W];
            const i32x4_t zrow = zrow_nx;
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
                const bool valid = (l0 + sc * SUB + i) < L;
                uu[i] = bio<T>::cv(ru[i]);
                float x = bio<T>::cv(rd[i]);
                if (DMODE != 2) x += bias;
                if (DMODE == 1) x = softplus_f(x);
                dl[i] = x;                                    // tail steps re-read row L-1: finite garbage that only meets g = 0
                zz[i] = HAS_Z ? bio<T>::cv(rz[i]) : 0.f;
                gg[i] = (valid && active) ? bio<T>::cv(rg[i]) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < H0W; ++k) ck_cur[k] = ck_nx[k];
            // ---- request the next one (ci - 1): lands while this one computes
            zrow_nx = zrow_is;
            issue_sub(ci - 1);                  // (ci = 0: rows 0 and an out-of-range checkpoint -- harmless, and no branch)
            // Halves in sequence (round 6; HALVES): the lane walks its channel's states in NH groups of NPH pairs, one after the other,
            // over the SAME sub-chunk -- inputs, conversions, softplus', the stores and every per-channel scalar are issued once, the
            // recomputed states of only one group are live (hs: 32 instead of 64 registers), and the registers that frees hold the decay
            // factors a = exp2(A2 dl) of the three recomputed steps, which the reverse sweep then REUSES instead of evaluating them again
            // (28 -> 16 v_exp_f32 and 6 packed multiplies less per wave-step).  The per-step sums over the states (G.B, A.Gt, C.h) are
            // carried from group to group in GB2s / dlA2s / yp2s; the step's outputs are formed and stored in the last group's sweep.
            f32x2 GB2s[SUB], dlA2s[SUB], yp2s[HAS_Z ? SUB : 1];
#pragma unroll
            for (int hf = 0; hf < NH; ++hf) {
            // one forward step of the group: h <- a*h + B*dl*u   (a kept in `keep` when the sweep will reuse it)
            const lds_cfptr bc_rec = lds_opaque(bc_chunk);
            float Brn[2 * NPH];                                          // LDS_AHEAD: the B row of the NEXT recomputed step
            if constexpr (LDS_AHEAD_REC) lds_ld_vec<2 * NPH>(Brn, bc_rec + ((sc * SUB) * 2 * N + hf * 2 * NPH));
            auto fwd_step = [&](f32x2(&h)[NPH], int i, f32x2(&keep)[NPH]) {
                float Bv[2 * NPH];
                const lds_cfptr brow = bc_rec + ((sc * SUB + i) * 2 * N + hf * 2 * NPH);          // re-read, do not keep rows in VGPRs
                if (DM_K2_EXP & 16) { for (int k = 0; k < 2 * NPH; ++k) Bv[k] = opaque(1.0f); } else if constexpr (LDS_AHEAD_REC) {
#pragma unroll
                    for (int k = 0; k < 2 * NPH; ++k) Bv[k] = Brn[k];
                    if (i + 1 < SUB - 1) lds_ld_vec<2 * NPH>(Brn, brow + 2 * N);
                    __builtin_amdgcn_sched_barrier(0);
                } else
                lds_ld_vec<2 * NPH>(Bv, brow);
                const float dlo = CACHE_A ? dl[i] : opaque(dl[i]);
                const float du = dlo * uu[i];
                float a_sh = 0.f;
                if (ASH) a_sh = fast_exp2(A2[0].x * dlo);             // DM_FLAG_A_SHARED: one decay factor for all states of the channel
#pragma unroll
                for (int k = 0; k < NPH; ++k) {
                    f32x2 a;
                    if (ASH) {
                        a = (f32x2){a_sh, a_sh};
                    } else {
                        const f32x2 t = A2[hf * NPH + k] * dlo;
                        a.x = fast_exp2(t.x);
                        a.y = fast_exp2(t.y);
                    }
                    if (CACHE_A) keep[k] = a;
                    f32x2 bb;
                    bb.x = Bv[2 * k];
                    bb.y = Bv[2 * k + 1];
                    h[k] = a * h[k] + bb * du;
                }
            };
            f32x2 h[NPH];
            unpack_state(h, ck_cur, hf);
            f32x2 hs[SUB][NPH];                                          // hs[i] = state before step sc*SUB+i
            f32x2 aa[CACHE_A ? SUB - 1 : 1][NPH];                        // decay factors of the recomputed steps
#pragma unroll
            for (int i = 0; i < SUB; ++i) {
#pragma unroll
                for (int k = 0; k < NPH; ++k) hs[i][k] = h[k];
                if (i < SUB - 1) fwd_step(h, i, aa[CACHE_A ? i : 0]);
            }
            // the state after the sub-chunk's last step is the checkpoint of the sub-chunk processed before: one recomputed step less
            if constexpr (HALVES) {
#pragma unroll
                for (int k = 0; k < NPH; ++k) { h[k] = hend[hf][k]; hend[hf][k] = hs[0][k]; }
            } else {
                unpack_state(h, ck_end, hf);
            }
            // ---- reverse sweep over the sub-chunk (h = state AFTER step j at the top of iteration j) -------
            // LDS_AHEAD: the B / C rows of step i - 1 are requested BEFORE step i is computed (a scheduling fence keeps the reads there),
            // so that no step starts by waiting for its own rows -- with two groups per step the sweep has twice as many read groups as
            // the whole-channel form, and SQ_WAIT_ANY had gone from 14 % to 23 % of the wave-cycles (profiles/r06_k2_decay_reuse.txt)
            const lds_cfptr bc_swp = lds_opaque(bc_chunk);
            float Bn[2 * NPH], Cn[2 * NPH];
            if constexpr (LDS_AHEAD) {
                const lds_cfptr r0 = bc_swp + ((sc * SUB + SUB - 1) * 2 * N + hf * 2 * NPH);
                lds_ld_vec<2 * NPH>(Bn, r0);
                lds_ld_vec<2 * NPH>(Cn, r0 + N);
            }
#pragma unroll
            for (int i = SUB - 1; i >= 0; --i) {
                const int j = sc * SUB + i;
                const int lraw = l0 + j;
                const bool valid = lraw < L;                    // wave-uniform
                const int l = valid ? lraw : L - 1;
                float Bv[2 * NPH], Cv[2 * NPH];
                const lds_cfptr brow = bc_swp + (j * 2 * N + hf * 2 * NPH);
                if (DM_K2_EXP & 16) { for (int k = 0; k < 2 * NPH; ++k) { Bv[k] = opaque(1.0f); Cv[k] = opaque(0.5f); } } else if constexpr (LDS_AHEAD) {
#pragma unroll
                    for (int k = 0; k < 2 * NPH; ++k) { Bv[k] = Bn[k]; Cv[k] = Cn[k]; }
                    if (i > 0) {
                        lds_ld_vec<2 * NPH>(Bn, brow - 2 * N);
                        lds_ld_vec<2 * NPH>(Cn, brow - 2 * N + N);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                lds_ld_vec<2 * NPH>(Bv, brow);
                lds_ld_vec<2 * NPH>(Cv, brow + N); }
                const float g = gg[i];
                float sz = 1.f, gy = g;
                if (HAS_Z) {
                    sz = sigmoid_f(zz[i]);
                    gy = g * zz[i] * sz;
                }
                const float dlo = (CACHE_A && i < SUB - 1) ? dl[i] : opaque(dl[i]);
                const float du = dlo * uu[i];
                float a_rev = 0.f;
                if (ASH) a_rev = fast_exp2(A2[0].x * dlo);
                f32x2 yp2 = (f32x2){0.f, 0.f}, GB2 = (f32x2){0.f, 0.f}, dlA2 = (f32x2){0.f, 0.f};
                if (hf > 0) { GB2 = GB2s[i]; dlA2 = dlA2s[i]; if (HAS_Z) yp2 = yp2s[i]; }
                float red[NH == 1 ? M : 1];
                uint32_t pk_all[M / 2 / NH];                   // bf16 pairs: [dB pairs (NPH) | dC pairs (NPH)] of this group
#pragma unroll
                for (int k = 0; k < NPH; ++k) {
                    const int kk = hf * NPH + k;               // the pair's place among the lane's NPL pairs
                    f32x2 bb, cc;
                    bb.x = Bv[2 * k]; bb.y = Bv[2 * k + 1];
                    cc.x = Cv[2 * k]; cc.y = Cv[2 * k + 1];
                    f32x2 a;
                    if (ASH) {
                        a = (f32x2){a_rev, a_rev};
                    } else if (CACHE_A && i < SUB - 1) {
                        a = aa[CACHE_A ? i : 0][k];
                    } else {
                        const f32x2 t = A2[kk] * dlo;
                        a.x = fast_exp2(t.x);
                        a.y = fast_exp2(t.y);
                    }
                    const f32x2 hj = h[k];
                    const f32x2 hp = hs[i][k];
                    if (HAS_Z) yp2 += cc * hj;
                    const f32x2 G = cc * gy + carry[kk];         // dL/dh_j
                    const f32x2 dCp = hj * gy;
                    carry[kk] = a * G;                           // a_j * dL/dh_j, flows to step j-1
                    const f32x2 Gt = carry[kk] * hp;             // = G * a * h_{j-1}
                    dlA2 += A2[kk] * Gt;
                    dA[kk] += Gt * dlo;
                    dA[kk].x = opaque(dA[kk].x);                 // accumulate NOW: left alone the scheduler defers all 8 steps'
                    dA[kk].y = opaque(dA[kk].y);                 // products to the chunk end and keeps 64 VGPRs alive for them
                    GB2 += G * bb;
                    const f32x2 dBp = G * du;
                    if constexpr (MFMA_RED) {
                        pk_all[k] = pack_bf16(dBp.x, dBp.y);
                        pk_all[NPH + k] = pack_bf16(dCp.x, dCp.y);
                    } else {
                        red[2 * k] = dBp.x;
                        red[2 * k + 1] = dBp.y;
                        red[NS + 2 * k] = dCp.x;
                        red[NS + 2 * k + 1] = dCp.y;
                    }
                    h[k] = hp;
                }
                if (hf < NH - 1) { GB2s[i] = GB2; dlA2s[i] = dlA2; if (HAS_Z) yp2s[i] = yp2; }
                if (hf == NH - 1) {
                const float ypre = slice_sum<SPLIT>(yp2.x + yp2.y) + Dv * uu[i];
                const float GB = slice_sum<SPLIT>(GB2.x + GB2.y);
                const float dlA = slice_sum<SPLIT>(dlA2.x + dlA2.y);
                float ddl = uu[i] * GB + LN2 * dlA;
                const float duv = dlo * GB + gy * Dv;
                if (DMODE != 0) ddl *= (1.0f - fast_exp2(-dlo * LOG2E));   // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
                if (q == 0) {                                             // one lane per channel owns the channel sums
                    dD_acc += gy * uu[i];
                    dbias_acc += ddl;
                }
                {
                    const int vo_s = valid ? vo_st : BIO_OOB;
                    if (!(DM_K2_EXP & 8) || duv == 123.456f) bio<T>::st_cv(r_du, vo_s, l * sl_du, duv);
                    if (!(DM_K2_EXP & 8) || ddl == 123.456f) bio<T>::st_cv(r_ddt, vo_s, l * sl_ddt, ddl);
                    if (HAS_Z) {
                        const float dzv = g * ypre * sz * (1.0f + zz[i] * (1.0f - sz));
                        bio<T>::st_cv(r_dz, vo_s, zrow[i] * sl_dz, dzv);
                    }
                }
                }
                if constexpr (DM_K2_EXP & 1) {
                } else if constexpr ((DM_K2_EXP & 32) != 0 && MFMA_RED) {       // products + conversions only: no MFMA, no LDS write (flush: bit 2)
#pragma unroll
                    for (int k = 0; k < M / 2 / NH; ++k) asm volatile("" ::"v"(pk_all[k]));
                } else if constexpr (MFMA_RED && NH == 2) {
                    // this group's 16 values (dB and dC of its 8 states) = ONE pair of MFMAs; LDS group hf (flush_dbc maps the columns)
                    const u32x4_t lo = {pk_all[0], pk_all[1], pk_all[2], pk_all[3]};
                    const u32x4_t hi = {pk_all[4], pk_all[5], pk_all[6], pk_all[7]};
                    f32x4 dsum;
                    if constexpr (DM_K2_EXP & 128) dsum = (f32x4){__uint_as_float(lo[0] ^ hi[0]), __uint_as_float(lo[1] ^ hi[1]), __uint_as_float(lo[2] ^ hi[2]), __uint_as_float(lo[3] ^ hi[3])};
                    else dsum = mfma_group_sum16(sel_lo, sel_hi, lo, hi);
                    if constexpr (DEFER_WRITE) {
                        // the totals leave for LDS one group-step LATER: written right behind its MFMAs the ds_write parks the wave until the
                        // matrix pipe has delivered (the read-ahead fence keeps the MFMAs at the end of the step)
                        if (pend_have) *reinterpret_cast<f32x4*>(&red_lds[wave][0][0] + pend_off) = pend;
                        pend = dsum;
                        pend_off = j * RED_ROW + hf * RED_HALF + red_slot;
                        pend_have = true;
                    } else {
                        *reinterpret_cast<f32x4*>(&red_lds[wave][j][hf * RED_HALF + red_slot]) = dsum;
                    }
                } else if constexpr (MFMA_RED) {
#pragma unroll
                    for (int g16 = 0; g16 < M / 16; ++g16) {   // 16 values (8 pairs) per pair of MFMAs; register r of group g16 = value 16*g16 + 4*(lane>>4) + r
                        const u32x4_t lo = {pk_all[8 * g16], pk_all[8 * g16 + 1], pk_all[8 * g16 + 2], pk_all[8 * g16 + 3]};
                        const u32x4_t hi = {pk_all[8 * g16 + 4], pk_all[8 * g16 + 5], pk_all[8 * g16 + 6], pk_all[8 * g16 + 7]};
                        const f32x4 dsum = mfma_group_sum16(sel_lo, sel_hi, lo, hi);
                        *reinterpret_cast<f32x4*>(&red_lds[wave][j][g16 * RED_HALF + red_slot]) = dsum;
                    }
                } else {
                    lane_group_reduce<M>(red);
#pragma unroll
                    for (int r4 = 0; r4 < R / 4; ++r4)
                        *reinterpret_cast<f32x4*>(&red_lds[wave][j][r4 * RED_HALF + red_slot]) = (f32x4){red[4 * r4], red[4 * r4 + 1], red[4 * r4 + 2], red[4 * r4 + 3]};
                }
            }
            }   // hf
#pragma unroll
            for (int k = 0; k < H0W; ++k) ck_end[k] = ck_cur[k];      // (dead in the HALVES form: removed by the compiler)
        }
        if (pend_have) *reinterpret_cast<f32x4*>(&red_lds[wave][0][0] + pend_off) = pend;
        if (!(DM_K2_EXP & 2)) __syncthreads();
        if (!(DM_K2_EXP & 35)) flush_dbc(ch);
        if (ch > 0) stash_bc(buf ^ 1, bc_next);
        if (!(DM_K2_EXP & 2)) __syncthreads();
        buf ^= 1;
    }
    if (active) {
        const int64_t pss_a = p.part_ss ? p.part_ss : (int64_t)p.dim * N, pss_d = p.part_ss ? p.part_ss : (int64_t)p.dim;
        float* dAp = p.dA_partial + (int64_t)s * pss_a + (int64_t)d * N + q * NS;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            dAp[2 * k] = dA[k].x;
            dAp[2 * k + 1] = dA[k].y;
        }
        if (q == 0) {
            if (p.dD_partial) p.dD_partial[(int64_t)s * pss_d + d] = dD_acc;
            if (p.dbias_partial) p.dbias_partial[(int64_t)s * pss_d + d] = dbias_acc;
        }
    }
}

// lanes per channel: a whole channel per lane up to d_state 16 (two lanes per channel = 3 waves per SIMD measured +10 % at d_state 16,
// profiles/r03_k2_experiments.txt), half of one at d_state 32
#ifndef DM_K2_SPLIT16
#define DM_K2_SPLIT16 1       // developer A/B: lanes per channel at d_state 16 (2 = half a channel per lane, 3 waves per SIMD)
#endif
template <int N> struct bwd_split { static constexpr int value = (N >= 32) ? 2 : (N == 16 ? DM_K2_SPLIT16 : 1); };

template <typename T, typename TBC, int N, bool HAS_Z, bool IDX>
static void launch_bwd2(const dm_scan_bwd_args& a, hipStream_t st, dim3 grid) {
    if constexpr (N == 16 && HAS_Z && IDX) {          // the one-exp variant is built for the Mamba-2 call pattern only
        if ((a.flags & DM_FLAG_A_SHARED) && (a.flags & DM_FLAG_DELTA_SOFTPLUS)) {
            hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, true, true, 1, true>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
            return;
        }
    }
    if constexpr (N == 16 && !HAS_Z && IDX) {         // the hoisted-gate / hoisted-softplus call pattern of the DiffMa mixer
        if (a.flags & DM_FLAG_DELTA_ACTIVATED) {
            hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, false, true, 2>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
            return;
        }
    }
    if (a.flags & DM_FLAG_DELTA_SOFTPLUS)
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, HAS_Z, IDX, 1>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
    else
        hipLaunchKernelGGL((scan_bwd_kernel<T, TBC, N, bwd_split<N>::value, HAS_Z, IDX, 0>), grid, dim3(WAVE * BWD_WAVES), 0, st, a);
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
