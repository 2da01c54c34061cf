// K6b  dm_ssd_bwd -- backward of the Mamba-2 single-chunk SSD core on the matrix pipe (twin of ssd.hip).
//
// Operator (reference call block/mamba2.py:392-410 under autograd), per (sequence, head), L <= 196, P = 64, N = 16:
//     dt = softplus(raw + bias),  s_l = A * cumsum(dt)_l,  M[l,i] = exp(s_l - s_i) [i <= l],  G = C B^T,  W = G .* M .* dt_i
//     Y = W X,   u = Y + D X,   out = u .* silu(z)
// Gradients, with gY = dout .* silu(z) and R = gY X^T (K = headdim):
//     dz = dout .* u .* silu'(z)                        dX = W^T gY + D gY                 dD = sum gY .* X
//     dG = R .* M .* dt_i:   dC = dG B,  dB = dG^T C    (partial per head; B and C are shared by the heads)
//     V  = R .* G .* M:      ds_l = sum_i V[l,i] dt_i - dt_l sum_q V[q,l];   d dt_i = sum_l V[l,i] + A * sum_{l >= i} ds_l
//     dA = sum_l ds_l cumsum(dt)_l;    d raw = d dt * sigmoid(raw + bias)
// Nothing is saved by the forward.  One 256-thread workgroup per (sequence, head) holds X, gY, B, C row-major in 79 KB of LDS
// (TWO workgroups per CU: the load / compute / store phases of different heads overlap) and walks the 28 causal 32 x 32 tile
// pairs twice, concurrently:
//   * waves 0-1, "T" orientation (score tile with keys as rows, queries as columns): its accumulator registers are the
//     A-operand of  Y += W X  and  dC += dG B  -- they own query tiles {6,3,2,0} and {5,4,1} (15 / 13 pairs);
//   * waves 2-3, "N" orientation (queries as rows, keys as columns): accumulators are the A-operand of  dX += W^T gY  and
//     dB += dG^T C  -- they own key tiles {0,3,4,6} and {1,2,5}.
// Each orientation computes its own G (1 MFMA) and R (4 MFMAs, K = 64) per pair, applies the decay (factorised per tile pair as in
// the forward: alpha_l * delta(lt,it) * gamma_i off the diagonal, element-wise exp + causal mask on the 7 diagonal tiles), feeds
// the rounded tiles straight back (no LDS round trip), and accumulates the row sums (T) / column sums (N) of V that d dt and dA
// need.  The second products contract over sequence positions, i.e. they want X / gY / B / C with positions in the K slots: instead
// of transposed LDS copies, the row-major fragments already loaded for R (rows = positions, K = channels) are multiplied by 0/1
// selector fragments -- the result X[position][channel] comes out in accumulator layout, which IS that B-operand layout (exact:
// one product by 1.0 per element).  16 MFMAs and ~160 VALU per pair and orientation.  Tile epilogues turn 8 rows at a time through a 2 KB staging tile so
// every global access is a 16-byte piece of a row.  The final d dt / dA reverse cumulative sum runs on the whole workgroup
// (shuffle scans inside the waves, one barrier).  Measured and profiled: DESIGN.md section 3 (K6b), profiles/r02_pmc_scan_kernels.txt.
#include "dm_common.h"
#include "ssd_common.h"

namespace dm {

constexpr int SB_MAXL = 196;                      // longest sequence
constexpr int SB_TILE = 32, SB_MAXT = 7;
constexpr int SB_TAB = SB_TILE * SB_MAXT;         // 224 tile-rounded positions
constexpr int SB_ROWS = SB_MAXL + 1;              // LDS rows of the operand arrays: row 196 is all zeros, every position past it aliases to it
constexpr int SB_WAVES = 4;
constexpr int SB_THREADS = 64 * SB_WAVES;
constexpr int SB_STG = 68;                        // dwords per staging-tile row (64 + 4)

// LDS map (byte offsets); rows [L, 196] of the operand arrays are zero
constexpr int SB_XS = 0;                          // X   [197][64]  16-byte chunks XOR-swizzled by (row & 7)
constexpr int SB_GS = SB_XS + SB_ROWS * 128;      // gY  [197][64]  same layout
constexpr int SB_BS = SB_GS + SB_ROWS * 128;      // B   [197][16]
constexpr int SB_CS = SB_BS + SB_ROWS * 32;       // C   [197][16]
constexpr int SB_TABS = SB_CS + SB_ROWS * 32;     // fp32 tables [T_NTAB][224]
enum { T_DT, T_CUM, T_S2, T_ALPHA, T_GAM, T_GDT, T_SIG, T_RS, T_CS, T_NTAB };
constexpr int SB_IDX = SB_TABS + T_NTAB * SB_TAB * 4;     // uint16 tables: z rows, dout rows
constexpr int SB_STAGE = SB_IDX + 2 * SB_TAB * 2;         // 4 waves x [8][SB_STG] fp32
constexpr int SB_RED = SB_STAGE + SB_WAVES * 8 * SB_STG * 4;   // block-reduction scratch
constexpr int SB_LDS_BYTES = SB_RED + 128;
static_assert(SB_LDS_BYTES <= 80 * 1024, "two workgroups per CU");
static_assert(SB_TABS % 16 == 0 && SB_STAGE % 16 == 0, "16-byte aligned LDS arrays");

__device__ __forceinline__ void wave_lds_sync() {         // make one wave's LDS writes visible to its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8 consecutive columns (chunk) of a row of the swizzled [197][64] arrays, as an MFMA operand fragment (rows >= 196: the zero row)
__device__ __forceinline__ ssd_u32x4 row_frag(const uint8_t* base, int row, int chunk) {
    const int rc = row < SB_MAXL ? row : SB_MAXL;
    return *reinterpret_cast<const ssd_u32x4*>(base + rc * 128 + ((chunk ^ (rc & 7)) << 4));
}
// 8 states of a B / C row
__device__ __forceinline__ ssd_u32x4 bc_frag(const uint8_t* base, int row, int kh) {
    const int rc = row < SB_MAXL ? row : SB_MAXL;
    return *reinterpret_cast<const ssd_u32x4*>(base + rc * 32 + kh * 16);
}
// One-hot selector fragment: slot e (0..7) of this lane holds 1.0, every other slot 0 (e outside 0..7: all zero)
template <typename T>
__device__ __forceinline__ ssd_u32x4 one_hot(int e) {
    constexpr uint32_t ONE = std::is_same<T, bf16_t>::value ? 0x3F80u : 0x3C00u;
    const uint32_t v = (e & 1) ? (ONE << 16) : ONE;
    const int d = e >> 1;
    const bool ok = e >= 0 && e < 8;
    return (ssd_u32x4){(ok && d == 0) ? v : 0u, (ok && d == 1) ? v : 0u, (ok && d == 2) ? v : 0u, (ok && d == 3) ? v : 0u};
}
// accumulator tile -> the two K-step fragments (8 positions each, accumulator order) of a 16-bit B-operand
template <typename O>
__device__ __forceinline__ void acc_to_frags(const f32x16& d, ssd_u32x4 (&f)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        f[ks] = (ssd_u32x4){O::pack(d[8 * ks], d[8 * ks + 1]), O::pack(d[8 * ks + 2], d[8 * ks + 3]), O::pack(d[8 * ks + 4], d[8 * ks + 5]),
                            O::pack(d[8 * ks + 6], d[8 * ks + 7])};
}

// Inclusive prefix sum over the workgroup's 256 threads (one value each): six shuffle steps inside each wave, one barrier to pass
// the wave totals on.  Every thread of the workgroup must call it.
__device__ __forceinline__ float block_prefix_sum(float v, int lane, int w, float* wtot) {
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const float t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    if (lane == WAVE - 1) wtot[w] = v;
    __syncthreads();
    float base = 0.0f;
#pragma unroll
    for (int j = 0; j < SB_WAVES - 1; ++j) base += (j < w) ? wtot[j] : 0.0f;
    return v + base;
}

template <typename T>
__global__ __launch_bounds__(SB_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void ssd_bwd_kernel(const dm_ssd_bwd_args p) {
    using O = ssd_ops<T>;
    constexpr int ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* const tab = reinterpret_cast<float*>(lds + SB_TABS);
    float* const DT = tab + T_DT * SB_TAB;
    float* const CUM = tab + T_CUM * SB_TAB;
    float* const S2 = tab + T_S2 * SB_TAB;
    float* const ALPHA = tab + T_ALPHA * SB_TAB;
    float* const GAM = tab + T_GAM * SB_TAB;
    float* const GDT = tab + T_GDT * SB_TAB;
    float* const SIG = tab + T_SIG * SB_TAB;
    float* const RS = tab + T_RS * SB_TAB;
    float* const CSV = tab + T_CS * SB_TAB;
    uint16_t* const zi = reinterpret_cast<uint16_t*>(lds + SB_IDX);
    uint16_t* const oi = zi + SB_TAB;
    float* const red = reinterpret_cast<float*>(lds + SB_RED);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.x, s = blockIdx.y;
    const int L = p.seqlen, H = p.nheads;
    const int bpd = (p.batch_per_dir > 0) ? p.batch_per_dir : p.nseq;
    const int dir = s / bpd;
    const int sb = s - dir * bpd;
    const int nt_l = (L + SB_TILE - 1) / SB_TILE;
    const int32_t* __restrict__ zidx = p.z_row_index ? p.z_row_index + (int64_t)dir * L : nullptr;
    const int32_t* __restrict__ oidx = p.out_row_index ? p.out_row_index + (int64_t)dir * L : nullptr;
    const float Ah = p.A[h], a2 = Ah * LOG2E, Dh = p.D ? p.D[h] : 0.0f, bias = p.dt_bias ? p.dt_bias[h] : 0.0f;
    const rsrc_t r_x = make_rsrc((const T*)p.x + (int64_t)s * p.x_ss);
    const rsrc_t r_B = make_rsrc((const T*)p.B + (int64_t)s * p.B_ss);
    const rsrc_t r_C = make_rsrc((const T*)p.C + (int64_t)s * p.C_ss);
    const rsrc_t r_z = make_rsrc(p.z ? (const T*)p.z + (int64_t)sb * p.z_ss : nullptr);
    const rsrc_t r_do = make_rsrc((const T*)p.dout + (int64_t)s * p.do_ss);
    const rsrc_t r_dx = make_rsrc((T*)p.dx + (int64_t)s * p.dx_ss);
    const rsrc_t r_dz = make_rsrc(p.dz ? (T*)p.dz + (int64_t)s * p.dz_ss : nullptr);
    const T* __restrict__ dtp = (const T*)p.dt + (int64_t)sb * p.dt_sb + h;
    const int sl_x = (int)p.x_sl * ES, sl_B = (int)p.B_sl * ES, sl_C = (int)p.C_sl * ES, sl_z = (int)p.z_sl * ES;
    const int sl_do = (int)p.do_sl * ES, sl_dx = (int)p.dx_sl * ES, sl_dz = (int)p.dz_sl * ES;
    const int hb = h * 64 * ES;                                                   // byte offset of the head's columns in a row
    float* const dbc_part = p.dBC_part + ((int64_t)h * p.nseq + s) * L * 32;          // head-major: the caller's sum over heads is a column sum

    // ---- phases 0 + 1 (round 6): EVERY global load of the (sequence, head) is requested up front, in two dependent waves ----------
    // Until round 6 the kernel walked a chain of ~6 memory latencies before its first MFMA (row indices -> dt -> [prefix sums] ->
    // x / dout / z of chunk w -> the same of chunk w + 4 -> B / C half 0 -> half 1), each ended by LDS stores or a barrier, with only
    // two workgroups per CU to cover for one another: 35 % of the wave-cycles parked at barriers, no unit busy
    // (profiles/r05_m2_ssd_pmc.txt).  Now: level 1 = B / C, the x rows and the row-index entries (the thread's own position and the
    // 4 rows the lane moves), level 2 = dt, dout, z through those indices -- 2 latencies; the tables are computed while level 2 flies.
    // phase 1 role: wave w moves the 16-byte chunks w and w + 4 of every row; lane rows r = lane + 64 j
    float dtv = 0.0f, sg0 = 0.0f;
    int zr0 = 0, orow0 = 0;
    ssd_u32x4 bq[2], cq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                                 // B / C: two 16-byte chunks per row, two rows per thread
        const int e = tid + SB_THREADS * i;
        const int r = e >> 1, hf = e & 1;
        const int rc = r < L ? r : L - 1;
        const auto b = __builtin_amdgcn_raw_buffer_load_b128(r_B, rc * sl_B + hf * 16, 0, 0);
        const auto c = __builtin_amdgcn_raw_buffer_load_b128(r_C, rc * sl_C + hf * 16, 0, 0);
        bq[i] = (ssd_u32x4){b[0], b[1], b[2], b[3]};
        cq[i] = (ssd_u32x4){c[0], c[1], c[2], c[3]};
    }
    const int pc = tid < L ? tid : L - 1;
    if (zidx) { zr0 = zidx[pc]; orow0 = oidx[pc]; } else { zr0 = pc; orow0 = pc; }
    int zrow[4], drow[4];
    ssd_u32x4 xq4[2][4], dq4[2][4], zq4[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = lane + 64 * j;
        const int rc = r < L ? r : L - 1;
        zrow[j] = zidx ? zidx[rc] : rc;
        drow[j] = oidx ? oidx[rc] : rc;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const auto xq = __builtin_amdgcn_raw_buffer_load_b128(r_x, rc * sl_x + hb + (w + 4 * m) * 16, 0, 0);
            xq4[m][j] = (ssd_u32x4){xq[0], xq[1], xq[2], xq[3]};
        }
    }
    // level 2: through the indices
    float rawv = 0.0f;
    if (tid < L) rawv = io<T>::ld(dtp + (int64_t)zr0 * p.dt_sl);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int cb = hb + (w + 4 * m) * 16;
            const auto dq = __builtin_amdgcn_raw_buffer_load_b128(r_do, drow[j] * sl_do + cb, 0, 0);
            dq4[m][j] = (ssd_u32x4){dq[0], dq[1], dq[2], dq[3]};
            zq4[m][j] = (ssd_u32x4){0u, 0u, 0u, 0u};
            if (p.z) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(r_z, zrow[j] * sl_z + cb, 0, 0);
                zq4[m][j] = (ssd_u32x4){v[0], v[1], v[2], v[3]};
            }
        }
    // B / C into LDS (level 1 has landed by the time the dt value is needed)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + SB_THREADS * i;
        const int r = e >> 1, hf = e & 1;
        if (r < SB_ROWS) {
            const ssd_u32x4 zero4 = {0u, 0u, 0u, 0u};
            *reinterpret_cast<ssd_u32x4*>(lds + SB_BS + r * 32 + hf * 16) = r < L ? bq[i] : zero4;
            *reinterpret_cast<ssd_u32x4*>(lds + SB_CS + r * 32 + hf * 16) = r < L ? cq[i] : zero4;
        }
    }
    // per-position scalars (thread t <-> position t)
    if (tid < L) {
        const float raw = rawv + bias;
        dtv = softplus_f(raw);
        sg0 = (raw > 20.0f) ? 1.0f : sigmoid_f(raw);                               // d softplus / d raw (identity above 20)
    } else {
        zr0 = 0;
        orow0 = 0;
    }
    if (tid < SB_TAB) {
        DT[tid] = dtv;
        SIG[tid] = sg0;
        zi[tid] = (uint16_t)zr0;
        oi[tid] = (uint16_t)orow0;
    }
    {
        const float c = block_prefix_sum(dtv, lane, w, red);                       // cumsum(dt)
        if (tid < SB_TAB) {
            CUM[tid] = c;
            S2[tid] = a2 * c;                                                     // log2-domain log-decay
        }
    }
    __syncthreads();
    if (tid < SB_TAB) {
        const int t = tid >> 5;
        const float sv = S2[tid];
        const float m_t = t ? S2[SB_TILE * t - 1] : 0.0f, m_n = S2[SB_TILE * t + SB_TILE - 1];
        const float al = fast_exp2(sv - m_t), ga = fast_exp2(m_n - sv);
        ALPHA[tid] = al;
        GAM[tid] = ga;
        GDT[tid] = ga * DT[tid];
    }
    auto m_of = [&](int t) -> float { return t ? S2[SB_TILE * t - 1] : 0.0f; };   // log-decay just before tile t

    // X, gY = dout .* silu(z) into LDS
    float dD_acc = 0.0f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ck = w + 4 * m;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = lane + 64 * j;
            if (r < SB_ROWS) {
                ssd_u32x4 xv = {0u, 0u, 0u, 0u}, gv = {0u, 0u, 0u, 0u};
                if (r < L) {
                    const ssd_u32x4 xq = xq4[m][j], dq = dq4[m][j], zq = zq4[m][j];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float g0 = O::lo(dq[d]), g1 = O::hi(dq[d]);
                        if (p.z) {
                            g0 *= silu_f(O::lo(zq[d]));
                            g1 *= silu_f(O::hi(zq[d]));
                        }
                        dD_acc += g0 * O::lo(xq[d]) + g1 * O::hi(xq[d]);
                        xv[d] = xq[d];
                        gv[d] = O::pack(g0, g1);
                    }
                }
                const int so = r * 128 + ((ck ^ (r & 7)) << 4);
                *reinterpret_cast<ssd_u32x4*>(lds + SB_XS + so) = xv;
                *reinterpret_cast<ssd_u32x4*>(lds + SB_GS + so) = gv;
            }
        }
    }
    __syncthreads();

    // ---- phase 2: the two orientations of the 28 tile pairs ---------------------------------------------------------------------
    const int q = lane & 31, kh = lane >> 5;                                      // fragment role: row/column of a tile, K half
    const int row8 = lane >> 3, c8 = lane & 7;                                    // epilogue role: row of an 8-row slab, 16-byte chunk
    float* const stage = reinterpret_cast<float*>(lds + SB_STAGE) + w * 8 * SB_STG;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // selectors: a [positions x 64 channels] row-major fragment set times these = the 32-channel half nt in accumulator layout
    // (K-step j of the half: channel 32 nt + q sits in slot q - 16 j - 8 kh); states likewise (16 of them, columns 16-31 empty)
    const ssd_u32x4 selx[2] = {one_hot<T>(q - 8 * kh), one_hot<T>(q - 16 - 8 * kh)};
    const ssd_u32x4 sel16 = one_hot<T>(q < 16 ? q - 8 * kh : -1);
    // role of the wave: waves land on SIMD (w & 3); the T waves carry the heavier epilogue, so odd heads swap the roles and the two
    // workgroups a CU holds put one T and one N wave on every SIMD
    const int wr = (w + 2 * (int)(blockIdx.x & 1)) & 3;

    if (wr < 2) {
        // ===== T orientation: query tile lt; rows of the score tile = keys (registers), columns = queries (lanes) =====
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int lt = (wr == 0) ? (pass == 0 ? 6 : pass == 1 ? 3 : pass == 2 ? 2 : 0) : (pass == 0 ? 5 : pass == 1 ? 4 : pass == 2 ? 1 : -1);
            if (lt < 0 || lt >= nt_l) continue;
            const int ql = SB_TILE * lt + q;
            const ssd_u32x4 cB = bc_frag(lds + SB_CS, ql, kh);
            ssd_u32x4 gB[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) gB[ks] = row_frag(lds + SB_GS, ql, 2 * ks + kh);
            const float s2q = S2[ql], aq = ALPHA[ql], mlt = m_of(lt);
            ssd_u32x4 zq4[4], dq4[4];                                             // the epilogue's z / dout pieces, in flight under the products
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int l = SB_TILE * lt + 8 * r4 + row8;
                const int lc = l < L ? l : L - 1;
                zq4[r4] = (ssd_u32x4){0u, 0u, 0u, 0u};
                dq4[r4] = (ssd_u32x4){0u, 0u, 0u, 0u};
                if (p.dz) {
                    const auto zq = __builtin_amdgcn_raw_buffer_load_b128(r_z, (int)zi[lc] * sl_z + hb + c8 * 16, 0, 0);
                    const auto dq = __builtin_amdgcn_raw_buffer_load_b128(r_do, (int)oi[lc] * sl_do + hb + c8 * 16, 0, 0);
                    zq4[r4] = (ssd_u32x4){zq[0], zq[1], zq[2], zq[3]};
                    dq4[r4] = (ssd_u32x4){dq[0], dq[1], dq[2], dq[3]};
                }
            }
            f32x16 Y0 = zero16, Y1 = zero16, dC = zero16;
            float rs = 0.0f;
#pragma unroll 1
            for (int it = 0; it <= lt; ++it) {
                const int ki = SB_TILE * it + q;
                const ssd_u32x4 bA = bc_frag(lds + SB_BS, ki, kh);
                f32x16 g = O::mfma(bA, cB, zero16);                               // G[query][key]: keys 32it + 4kh + 8r4 + r in registers
                ssd_u32x4 xA[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) xA[ks] = row_frag(lds + SB_XS, ki, 2 * ks + kh);
                f32x16 rr = zero16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) rr = O::mfma(xA[ks], gB[ks], rr);   // R = gY . X
                // X and B of the key tile with the keys in the K slots (accumulator order): selector products
                ssd_u32x4 xf0[2], xf1[2], bf[2];
                acc_to_frags<O>(O::mfma(xA[1], selx[1], O::mfma(xA[0], selx[0], zero16)), xf0);
                acc_to_frags<O>(O::mfma(xA[3], selx[1], O::mfma(xA[2], selx[0], zero16)), xf1);
                acc_to_frags<O>(O::mfma(bA, sel16, zero16), bf);
                const int kb = SB_TILE * it + 4 * kh;
                float md[16];                                                     // M[query][key] * dt_key
                if (it < lt) {
                    const float cq = aq * fast_exp2(mlt - S2[SB_TILE * it + SB_TILE - 1]);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 f = *reinterpret_cast<const f32x4*>(&GDT[kb + 8 * r4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) md[4 * r4 + r] = cq * f[r];
                    }
                } else {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 sk = *reinterpret_cast<const f32x4*>(&S2[kb + 8 * r4]);
                        const f32x4 dk = *reinterpret_cast<const f32x4*>(&DT[kb + 8 * r4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) md[4 * r4 + r] = (kb + 8 * r4 + r <= ql) ? fast_exp2(s2q - sk[r]) * dk[r] : 0.0f;
                    }
                }
                uint32_t wp[8], dp[8];
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const float w0 = g[2 * r2] * md[2 * r2], w1 = g[2 * r2 + 1] * md[2 * r2 + 1];
                    rs += w0 * rr[2 * r2] + w1 * rr[2 * r2 + 1];                   // row sum of R .* W
                    wp[r2] = O::pack(w0, w1);
                    dp[r2] = O::pack(rr[2 * r2] * md[2 * r2], rr[2 * r2 + 1] * md[2 * r2 + 1]);
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const ssd_u32x4 wf = {wp[4 * ks], wp[4 * ks + 1], wp[4 * ks + 2], wp[4 * ks + 3]};
                    const ssd_u32x4 df = {dp[4 * ks], dp[4 * ks + 1], dp[4 * ks + 2], dp[4 * ks + 3]};
                    Y0 = O::mfma(wf, xf0[ks], Y0);
                    Y1 = O::mfma(wf, xf1[ks], Y1);
                    dC = O::mfma(df, bf[ks], dC);
                }
            }
            rs += __shfl_xor(rs, 32);
            if (kh == 0) RS[ql] = rs;
            if (q < 16) {                                                         // dC[32lt + 4kh + 8r4 + r][state q], partial of this head
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = SB_TILE * lt + 4 * kh + 8 * r4 + r;
                        if (row < L) dbc_part[row * 32 + 16 + q] = dC[4 * r4 + r];
                    }
            }
            // dz = dout .* (Y + D x) .* silu'(z): 8 rows at a time through the staging tile
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    stage[(4 * kh + r) * SB_STG + q] = Y0[4 * r4 + r];
                    stage[(4 * kh + r) * SB_STG + 32 + q] = Y1[4 * r4 + r];
                }
                wave_lds_sync();
                const f32x4 ua = *reinterpret_cast<const f32x4*>(&stage[row8 * SB_STG + 8 * c8]);
                const f32x4 ub = *reinterpret_cast<const f32x4*>(&stage[row8 * SB_STG + 8 * c8 + 4]);
                wave_lds_sync();
                const int l = SB_TILE * lt + 8 * r4 + row8;
                if (p.dz && l < L) {
                    const ssd_u32x4 xq = row_frag(lds + SB_XS, l, c8);
                    const int zr = zi[l];
                    const ssd_u32x4 zq = zq4[r4], dq = dq4[r4];
                    ssd_u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        float res[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const float u = (d < 2 ? ua[2 * d + e] : ub[2 * (d - 2) + e]) + Dh * (e ? O::hi(xq[d]) : O::lo(xq[d]));
                            const float zv = e ? O::hi(zq[d]) : O::lo(zq[d]);
                            const float dv = e ? O::hi(dq[d]) : O::lo(dq[d]);
                            const float sg = sigmoid_f(zv);
                            res[e] = dv * u * sg * (1.0f + zv * (1.0f - sg));
                        }
                        o[d] = O::pack(res[0], res[1]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, r_dz, zr * sl_dz + hb + c8 * 16, 0, 0);
                }
            }
        }
    } else {
        // ===== N orientation: key tile it; rows of the score tile = queries (registers), columns = keys (lanes) =====
        const int v = wr - 2;
#pragma unroll 1
        for (int pass = 0; pass < 4; ++pass) {
            const int it = (v == 0) ? (pass == 0 ? 0 : pass == 1 ? 3 : pass == 2 ? 4 : 6) : (pass == 0 ? 1 : pass == 1 ? 2 : pass == 2 ? 5 : -1);
            if (it < 0 || it >= nt_l) continue;
            const int ki = SB_TILE * it + q;
            const int kic = ki < SB_TAB ? ki : SB_TAB - 1;
            const ssd_u32x4 bB = bc_frag(lds + SB_BS, ki, kh);
            ssd_u32x4 xB[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) xB[ks] = row_frag(lds + SB_XS, ki, 2 * ks + kh);
            const float dtk = DT[kic], gamk = GAM[kic], s2k = S2[kic], mnext = S2[SB_TILE * it + SB_TILE - 1];
            f32x16 X0 = zero16, X1 = zero16, dB = zero16;
            float cv = 0.0f;
#pragma unroll 1
            for (int lt = it; lt < nt_l; ++lt) {
                const int qrow = SB_TILE * lt + q;
                const ssd_u32x4 cA = bc_frag(lds + SB_CS, qrow, kh);
                f32x16 g = O::mfma(cA, bB, zero16);                               // G[query][key]: queries 32lt + 4kh + 8r4 + r in registers
                ssd_u32x4 gA[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) gA[ks] = row_frag(lds + SB_GS, qrow, 2 * ks + kh);
                f32x16 rr = zero16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) rr = O::mfma(gA[ks], xB[ks], rr);
                ssd_u32x4 gf0[2], gf1[2], cf[2];                                   // gY and C of the query tile with the queries in the K slots
                acc_to_frags<O>(O::mfma(gA[1], selx[1], O::mfma(gA[0], selx[0], zero16)), gf0);
                acc_to_frags<O>(O::mfma(gA[3], selx[1], O::mfma(gA[2], selx[0], zero16)), gf1);
                acc_to_frags<O>(O::mfma(cA, sel16, zero16), cf);
                const int qb = SB_TILE * lt + 4 * kh;
                float mf[16];                                                     // M[query][key] (without dt)
                if (lt > it) {
                    const float ck = gamk * fast_exp2(m_of(lt) - mnext);
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 a = *reinterpret_cast<const f32x4*>(&ALPHA[qb + 8 * r4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) mf[4 * r4 + r] = ck * a[r];
                    }
                } else {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 sq = *reinterpret_cast<const f32x4*>(&S2[qb + 8 * r4]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) mf[4 * r4 + r] = (ki <= qb + 8 * r4 + r) ? fast_exp2(sq[r] - s2k) : 0.0f;
                    }
                }
                uint32_t wp[8], dp[8];
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const float gm0 = g[2 * r2] * mf[2 * r2], gm1 = g[2 * r2 + 1] * mf[2 * r2 + 1];
                    cv += gm0 * rr[2 * r2] + gm1 * rr[2 * r2 + 1];                 // column sum of V = R .* G .* M
                    wp[r2] = O::pack(gm0 * dtk, gm1 * dtk);
                    dp[r2] = O::pack(rr[2 * r2] * (mf[2 * r2] * dtk), rr[2 * r2 + 1] * (mf[2 * r2 + 1] * dtk));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const ssd_u32x4 wf = {wp[4 * ks], wp[4 * ks + 1], wp[4 * ks + 2], wp[4 * ks + 3]};
                    const ssd_u32x4 df = {dp[4 * ks], dp[4 * ks + 1], dp[4 * ks + 2], dp[4 * ks + 3]};
                    X0 = O::mfma(wf, gf0[ks], X0);
                    X1 = O::mfma(wf, gf1[ks], X1);
                    dB = O::mfma(df, cf[ks], dB);
                }
            }
            cv += __shfl_xor(cv, 32);
            if (kh == 0) CSV[ki] = cv;
            if (q < 16) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = SB_TILE * it + 4 * kh + 8 * r4 + r;
                        if (row < L) dbc_part[row * 32 + q] = dB[4 * r4 + r];
                    }
            }
            // dx = W^T gY + D gY
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    stage[(4 * kh + r) * SB_STG + q] = X0[4 * r4 + r];
                    stage[(4 * kh + r) * SB_STG + 32 + q] = X1[4 * r4 + r];
                }
                wave_lds_sync();
                const f32x4 ua = *reinterpret_cast<const f32x4*>(&stage[row8 * SB_STG + 8 * c8]);
                const f32x4 ub = *reinterpret_cast<const f32x4*>(&stage[row8 * SB_STG + 8 * c8 + 4]);
                wave_lds_sync();
                const int l = SB_TILE * it + 8 * r4 + row8;
                if (l < L) {
                    const ssd_u32x4 gq = row_frag(lds + SB_GS, l, c8);
                    ssd_u32x4 o;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const float u0 = (d < 2 ? ua[2 * d] : ub[2 * (d - 2)]) + Dh * O::lo(gq[d]);
                        const float u1 = (d < 2 ? ua[2 * d + 1] : ub[2 * (d - 2) + 1]) + Dh * O::hi(gq[d]);
                        o[d] = O::pack(u0, u1);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, r_dx, l * sl_dx + hb + c8 * 16, 0, 0);
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 3: d dt, dA, dD (thread t works on position 223 - t: the cumulative sum of d s runs from the end) ----------------
    const int pos = SB_TAB - 1 - tid;
    const bool live = tid < SB_TAB && pos < L;
    float csv = 0.0f, dsv = 0.0f;
    if (live) {
        csv = CSV[pos];
        dsv = RS[pos] - DT[pos] * csv;                                            // d s_l
    }
    const float rc = block_prefix_sum(dsv, lane, w, red + 24);                    // sum_{l >= pos} d s_l
    const float draw = live ? (csv + Ah * rc) * SIG[pos] : 0.0f;                  // d(raw dt) through softplus
    {
        const float a = wave_sum_dpp(live ? dsv * CUM[pos] : 0.0f), d = wave_sum_dpp(dD_acc), b = wave_sum_dpp(draw);
        if (lane == 0) {
            red[w] = a;
            red[8 + w] = d;
            red[16 + w] = b;
        }
    }
    if (live) p.ddt[((int64_t)s * L + zi[pos]) * H + h] = draw;
    __syncthreads();
    if (tid < 3) {                                                                // dA | dD | d dt_bias partial sums of this (sequence, head)
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < SB_WAVES; ++k) t += red[8 * tid + k];
        p.dAD_part[((int64_t)s * 3 + tid) * H + h] = t;
    }
}

}  // namespace dm

extern "C" int dm_ssd_bwd_supported(int seqlen, int headdim, int dstate, int io_dtype) {
    return (seqlen >= 1 && seqlen <= dm::SB_MAXL && headdim == 64 && dstate == 16 && (io_dtype == DM_BF16 || io_dtype == DM_F16)) ? 1 : 0;
}

extern "C" int dm_ssd_bwd(const dm_ssd_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_ssd_bwd: null args"); return DM_ERR_ARG; }
    const dm_ssd_bwd_args& a = *args;
    if (!a.x || !a.B || !a.C || !a.dt || !a.dout || !a.A || !a.dx || !a.dBC_part || !a.ddt || !a.dAD_part) {
        set_error("dm_ssd_bwd: null tensor pointer"); return DM_ERR_ARG;
    }
    if ((a.z == nullptr) != (a.dz == nullptr)) { set_error("dm_ssd_bwd: z and dz go together"); return DM_ERR_ARG; }
    if (a.nseq <= 0 || a.nheads <= 0 || a.seqlen <= 0) { set_error("dm_ssd_bwd: non-positive size"); return DM_ERR_ARG; }
    if (!dm_ssd_bwd_supported(a.seqlen, a.headdim, a.dstate, a.io_dtype)) {
        set_error("dm_ssd_bwd: needs 16-bit I/O, headdim 64, d_state 16, seqlen <= %d (got L %d P %d N %d dtype %d)", SB_MAXL, a.seqlen, a.headdim, a.dstate, a.io_dtype);
        return DM_ERR_ARG;
    }
    if (a.nseq > 65535) { set_error("dm_ssd_bwd: nseq %d > 65535", a.nseq); return DM_ERR_ARG; }
    if (a.batch_per_dir > 0 && a.nseq % a.batch_per_dir != 0) { set_error("dm_ssd_bwd: nseq %% batch_per_dir != 0"); return DM_ERR_ARG; }
    if ((a.z_row_index == nullptr) != (a.out_row_index == nullptr)) { set_error("dm_ssd_bwd: both row-index tables or neither"); return DM_ERR_ARG; }
    auto bad = [](const void* ptr, int64_t s0, int64_t s1) { return ((uintptr_t)ptr & 15) || (s0 & 7) || (s1 & 7); };
    if (bad(a.x, a.x_ss, a.x_sl) || bad(a.B, a.B_ss, a.B_sl) || bad(a.C, a.C_ss, a.C_sl) || bad(a.dout, a.do_ss, a.do_sl) ||
        bad(a.dx, a.dx_ss, a.dx_sl) || (a.z && (bad(a.z, a.z_ss, a.z_sl) || bad(a.dz, a.dz_ss, a.dz_sl)))) {
        set_error("dm_ssd_bwd: rows must be 16-byte aligned (tiles move as 16-byte pieces)"); return DM_ERR_LAYOUT;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(a.nheads, a.nseq), block(SB_THREADS);
    hipError_t e;
    // the dynamic-LDS limit of a kernel is a per-DEVICE attribute: set it once per (device, instantiation), not once per process
    // (a process that launches on a second GPU would otherwise fail there with 79 KB of dynamic LDS -- ADVICE r2)
    static bool reserved[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("dm_ssd_bwd: cannot query the current device"); return DM_ERR_LAUNCH; }
    const int which = a.io_dtype == DM_BF16 ? 0 : 1;
    if (!reserved[which][dev]) {
        const hipError_t r = which == 0 ? hipFuncSetAttribute((const void*)ssd_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS_BYTES)
                                        : hipFuncSetAttribute((const void*)ssd_bwd_kernel<f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, SB_LDS_BYTES);
        if (r != hipSuccess) { set_error("dm_ssd_bwd: cannot reserve %d bytes of LDS: %s", SB_LDS_BYTES, hipGetErrorString(r)); return DM_ERR_LAUNCH; }
        reserved[which][dev] = true;                  // benign race: two threads may both set the same value
    }
    if (which == 0) hipLaunchKernelGGL((ssd_bwd_kernel<bf16_t>), grid, block, SB_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((ssd_bwd_kernel<f16_t>), grid, block, SB_LDS_BYTES, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_ssd_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
