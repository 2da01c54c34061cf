// K12  dm_gemm_large -- C[P][Q] = A[P][Kc] * B[Q][Kc]^T for the LARGE-BATCH projections, 16-bit operands, fp32 accumulation, gfx950.
//
// The dense projections of the mixer at the bench batch (reference block/mamba.py:261 in_proj, :315 out_proj, called at :333-337):
// M = B L = 100 352 rows against K = 512 / 1024 / 2048 and N = 2048 / 512 / 1024 -- a very tall product with a SHORT contraction.
// What that shape asks of a kernel: a 256 x 256 output tile lives for only 8-16 K-steps of 64, so the start of a
// tile (first operand fetch) and its end (a 128 KB C tile) are a third of its life; a kernel that drains its load pipeline at every
// tile boundary, as one launch-per-tile-grid kernels do, cannot pass ~35 % of the matrix pipe here.  This one is PERSISTENT:
//   * one 512-thread workgroup per CU walks a list of output tiles; the operand stream (LDS-DMA, `buffer_load_dwordx4 ... lds`) runs
//     three K-steps ahead of the products ACROSS tile boundaries -- the next tile's first operands land while the current tile's
//     last products issue and its C tile leaves;
//   * a 4-stage LDS ring of (256 + 256) x 32 operand slices (128 KB), counted `s_waitcnt vmcnt(N)` (never 0 in the loop) and ONE
//     `s_barrier` per K-step; fragments for the next half-step are read while the current half-step's 8 MFMAs issue;
//   * v_mfma_f32_32x32x16 with the operands SWAPPED (weights as A, activations as B): a lane then holds 4 consecutive output
//     columns of one row; `v_permlane32_swap` pairs two lanes' quads into 16-byte pieces -> `buffer_store_dwordx4`;
//   * XCD-aware tile order: the 32 CUs of an XCD work on 32 CONSECUTIVE tiles (row-block major), so the column tiles of a row block
//     read its A rows from that XCD's L2 while the (small) weight matrix stays L2-resident everywhere.
// Both operands k-major ("NT"): the forward y = x W^T directly; the input gradient dx = dy W through a transposed 16-bit copy of
// the weight (2 MB: the caller makes it once per step).  The weight gradient (contraction over M) is a different kernel.
//
// LDS image of an operand slice: [256 rows][32 k] 16-bit = 64-byte rows, written by LDS-DMA in lane order (lane l of a wave
// instruction: row l >> 2, 16-byte slot l & 3 of a 16-row group) -- the swizzle is therefore applied to the SOURCE address: slot s of
// row r holds the row's piece s ^ ((r >> 2) & 3).  A fragment read (ds_read_b128: lane = row, all lanes one piece) then touches
// 16 distinct 16-byte slots of the 256-byte bank window per 16-lane group: conflict-free.
#include "dm_common.h"
#include <cstdlib>

namespace dm {

constexpr int GL_BM = 256, GL_BN = 256, GL_BK = 32, GL_NS = 4;
constexpr int GL_SLICE = GL_BM * GL_BK * 2;            // bytes of one operand slice (16 KB)
constexpr int GL_B0 = GL_NS * GL_SLICE;                // the B ring starts behind the A ring
constexpr int GL_RING = 2 * GL_NS * GL_SLICE;          // 128 KB of operand ring ...
constexpr int GL_LDS = GL_RING + 8 * 4096;             // ... + 4 KB per wave to turn the C tile's fragments into whole 128-byte rows
constexpr int GL_THREADS = 512;

struct gl_args {
    const void *a, *b;
    void* c;
    int M, N, K;
    int lda, ldb, ldc;           // elements
    int MB, NT, T, TX;           // row blocks, column tiles, tiles, tiles per XCD
};

typedef __bf16 gl_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gl_f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* gl_lds_ptr;

template <typename T> struct gl_mfma;
template <> struct gl_mfma<bf16_t> {
    static __device__ __forceinline__ f32x16 run(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gl_bf16x8, a), __builtin_bit_cast(gl_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return dm_cvt_pk_bf16(lo, hi); }
};
template <> struct gl_mfma<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x16 run(const u32x4_t& a, const u32x4_t& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gl_f16x8, a), __builtin_bit_cast(gl_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
};

__device__ __forceinline__ void gl_swap32(uint32_t& a, uint32_t& b) {        // a[lanes 32..63] <-> b[lanes 0..31]
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// The running position of the operand stream: which of this workgroup's tiles, which K-step.
struct gl_cursor {
    int tile, kt;                // index into the workgroup's own tile list, K-step inside the tile
    int soff_a, soff_b;          // byte offsets of the tile's first slice (rows m0 / n0, k = 0)
};

template <typename T, bool NT_STORE = true>
__global__ __launch_bounds__(GL_THREADS, 2) void gemm_large_kernel(const gl_args p) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[GL_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;                       // wave tile: rows 64 wm .. + 64, columns 128 wn .. + 128
    const int KT = p.K / GL_BK, KG = KT / 16;                      // K-steps of a tile; groups of 16 steps (the unrolled loop body)

    // ---- this workgroup's tiles: XCD x owns tiles [x TX, (x + 1) TX), its 32 workgroups take them 32 at a time ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int t0 = xcd * p.TX + slot;
    const int t_end = min((xcd + 1) * p.TX, p.T);
    const int ntile = t0 < t_end ? (t_end - t0 + 31) / 32 : 0;
    if (ntile == 0) return;
    const int G = ntile * KT;                                      // K-steps of this workgroup

    // buffer descriptors: rows past the matrix end read as zeros / are dropped on store
    const rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.b), 0, (int)(((int64_t)(p.N - 1) * p.ldb + p.K) * 2), 0x00020000);
    const int c_bytes = (int)(((int64_t)(p.M - 1) * p.ldc + p.N) * 2);

    // ---- operand stream: a wave issues 16-row groups 2 wave, 2 wave + 1 of both operands' slices (4 LDS-DMA instructions per K-step) ----
    const int lrow = lane >> 2, lpiece = (lane & 3) ^ ((lane >> 4) & 3);
    int voa[2], vob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * (2 * wave + j) + lrow;
        voa[j] = row * p.lda * 2 + lpiece * 16;
        vob[j] = row * p.ldb * 2 + lpiece * 16;
    }
    const int lds_w = wave * 2048;                                 // this wave's two 1 KB groups inside a slice

    auto tile_origin = [&](int ti, int& soa, int& sob, int& soc) {
        const int t = t0 + 32 * ti;
        const int mb = t / p.NT, nb = t - mb * p.NT;
        soa = mb * GL_BM * p.lda * 2;
        sob = nb * GL_BN * p.ldb * 2;
        soc = (mb * GL_BM * p.ldc + nb * GL_BN) * 2;
    };
    gl_cursor ld;
    ld.tile = 0;
    ld.kt = 0;
    int soc_dummy;
    tile_origin(0, ld.soff_a, ld.soff_b, soc_dummy);
    int lg = 0;                                                     // K-steps whose loads have been issued

    auto issue_loads = [&](int stage) {
        // past the end of the list: offsets beyond the descriptors (zeros land in LDS, nobody reads them); the loads are still
        // ISSUED so that the counted waits below always see the same queue
        const bool live = lg < G;
        const int sa = live ? ld.soff_a + ld.kt * (GL_BK * 2) : 0x7ffffff0;
        const int sb = live ? ld.soff_b + ld.kt * (GL_BK * 2) : 0x7ffffff0;
        uint8_t* la = lds + stage * GL_SLICE + lds_w;
        uint8_t* lb = lds + GL_B0 + stage * GL_SLICE + lds_w;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (gl_lds_ptr)(la), 16, voa[0], sa, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (gl_lds_ptr)(la + 1024), 16, voa[1], sa, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (gl_lds_ptr)(lb), 16, vob[0], sb, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (gl_lds_ptr)(lb + 1024), 16, vob[1], sb, 0, 0);
        ++lg;
        if (++ld.kt == KT) {
            ld.kt = 0;
            ++ld.tile;
            if (ld.tile < ntile) tile_origin(ld.tile, ld.soff_a, ld.soff_b, soc_dummy);
        }
    };

    // ---- fragment addresses: lane = row (l & 31), k-group g = l >> 5; half-step h reads piece 2 h + g of the 32-k slice.  The four
    // bases are made opaque so that the ring / fragment offsets stay instruction immediates (below 64 KB from each base) ----
    const int frow = lane & 31, fg = lane >> 5, fsw = (frow >> 2) & 3;
    int xa0 = (64 * wm + frow) * 64 + ((fg ^ fsw) << 4);                    // activations (MFMA B operand): 2 fragments of 32 rows
    int wa0 = GL_B0 + (128 * wn + frow) * 64 + ((fg ^ fsw) << 4);           // weights (MFMA A operand): 4 fragments
    int xa1 = xa0 ^ 32, wa1 = wa0 ^ 32;                                     // half-step 1: pieces 2 + g
    asm volatile("" : "+v"(xa0), "+v"(xa1), "+v"(wa0), "+v"(wa1));
    auto read_x = [&](int stage, int half, int i) -> u32x4_t {
        return *reinterpret_cast<const u32x4_t*>(lds + (half ? xa1 : xa0) + (stage * GL_SLICE + i * 2048));
    };
    auto read_w = [&](int stage, int half, int i) -> u32x4_t {
        return *reinterpret_cast<const u32x4_t*>(lds + (half ? wa1 : wa0) + (stage * GL_SLICE + i * 2048));
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- the finished C tile waits in 16 x 16-byte pieces per lane (the "stash") and leaves ONE piece per K-step of the next tile:
    // the write stream is even in time (a burst of 128 KB per CU at every tile end would queue in front of the operand loads, which
    // share the in-order vmcnt counter with it) ----
    u32x4_t stash[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) stash[i] = (u32x4_t){0u, 0u, 0u, 0u};
    int soc_st = 0, c_live = 0;                                    // C origin of the stashed tile; its descriptor size (0: nothing to store)
    // A lane holds, of the 32 x 32 block (i, j) of the wave's 64 x 128 tile, row 32 j + (l & 31), columns 32 i + 8 q + 4 (l >> 5) + 0..3:
    // stored like that, every store instruction would write 32-byte pieces of 32 different rows, and with the pieces of a row
    // leaving K-steps apart the L2 would hold (and evict) partially written lines.  So the tile is turned in LDS first, 32 rows x
    // 128 bytes at a time through the wave's private 4 KB: in as 16-byte fragments, out as rows -- a stash piece is then 8 rows x
    // 128 bytes, one whole cache line per row and store instruction.  16-byte slot s of staging row r lies at slot s ^ (r & 7).
    uint8_t* const tw = lds + GL_RING + wave * 4096;
    int tw_w = frow * 128, tw_r = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
    const int tw_x = frow & 7;
    asm volatile("" : "+v"(tw_w), "+v"(tw_r));
    const int voc = ((64 * wm + (lane >> 3)) * p.ldc + 128 * wn + 8 * (lane & 7)) * 2;
    auto to_stash = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const f32x16& v = acc[2 * h + ii][j];
                    uint32_t w[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w[2 * q] = gl_mfma<T>::pack(v[4 * q], v[4 * q + 1]);
                        w[2 * q + 1] = gl_mfma<T>::pack(v[4 * q + 2], v[4 * q + 3]);
                    }
                    // quads of lanes l and l + 32 are neighbours in the row: after the swaps a lane below 32 holds columns 0..7 and
                    // 16..23 of the 32-column block, its partner 8..15 and 24..31
                    gl_swap32(w[0], w[2]);
                    gl_swap32(w[1], w[3]);
                    gl_swap32(w[4], w[6]);
                    gl_swap32(w[5], w[7]);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int slot = ii * 4 + u * 2;                               // + (l >> 5)
                        *reinterpret_cast<u32x4_t*>(tw + tw_w + ((((slot + fg) ^ tw_x)) << 4)) = (u32x4_t){w[4 * u], w[4 * u + 1], w[4 * u + 2], w[4 * u + 3]};
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) stash[j * 8 + h * 4 + k] = *reinterpret_cast<const u32x4_t*>(tw + tw_r + k * 1024);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    auto store_piece = [&](int pc, int live_bytes) {             // piece pc = 8 j + 4 h + k: rows 32 j + 8 k + (l >> 3), columns 64 h + 8 (l & 7) ..
        const rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, live_bytes, 0x00020000);      // size 0: the store is dropped (but counted)
        const int j = pc >> 3, h = (pc >> 2) & 1, k = pc & 3;
        // aux 2 = nt: the tile is written once and never read here; measured 239-248 us against 292 with default-policy stores
        // (in_proj forward) -- a store is acknowledged sooner, and the operand loads queue behind the stores in vmcnt order
        __builtin_amdgcn_raw_buffer_store_b128(stash[pc], rc, voc + h * 128, soc_st + (32 * j + 8 * k) * p.ldc * 2, NT_STORE ? 2 : 0);
    };

    // ---- prologue: three K-steps in flight, the first two landed ----
    issue_loads(0);
    issue_loads(1);
    issue_loads(2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4_t fx0[2], fx1[2], fw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fw[i] = read_w(0, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) fx0[i] = read_x(0, 0, i);

    int soa_c, sob_c, soc_c;
    tile_origin(0, soa_c, sob_c, soc_c);

    // One half-step: 8 MFMAs on (fw, fxc); the X fragments of the NEXT half-step are requested first, every W fragment is re-read
    // (next half-step's) right behind its two products -- one W set in registers, the reads spread between the products.
#define GL_HALF(FXC, FXN, NSTAGE, NHALF)                                                                                     \
    {                                                                                                                        \
        FXN[0] = read_x(NSTAGE, NHALF, 0);                                                                                   \
        FXN[1] = read_x(NSTAGE, NHALF, 1);                                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                      \
            acc[i][0] = gl_mfma<T>::run(fw[i], FXC[0], acc[i][0]);                                                           \
            acc[i][1] = gl_mfma<T>::run(fw[i], FXC[1], acc[i][1]);                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
            fw[i] = read_w(NSTAGE, NHALF, i);                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                               \
        }                                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                            \
    }
    // One K-step at position J of a 16-step group (ring stage J & 3, stash piece J).  Queue behind the loads of K-step g + 2 when the
    // wait is reached: store (g - 1), 4 loads (g + 3), store (g) -> vmcnt(6), always (stores of an empty stash are issued and dropped).
#define GL_STEP(J)                                                                                                           \
    {                                                                                                                        \
        GL_HALF(fx0, fx1, (J) & 3, 1)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        issue_loads(((J) + 3) & 3);                                                                                          \
        store_piece(J, st_bytes);                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        GL_HALF(fx1, fx0, ((J) + 1) & 3, 0)                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
        if ((J) == 15 && last_group) {                                                                                       \
            to_stash();                                                                                                      \
            soc_st = soc_c;                                                                                                  \
            c_live = c_bytes;                                                                                                \
        }                                                                                                                    \
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                     \
        __builtin_amdgcn_s_barrier();                                                              \
    }

    for (int ct = 0; ct < ntile; ++ct) {
        tile_origin(ct, soa_c, sob_c, soc_c);
        for (int kg = 0; kg < KG; ++kg) {
            const int st_bytes = kg == 0 ? c_live : 0;            // the stash leaves during the first 16 steps of the next tile
            const bool last_group = kg == KG - 1;
            GL_STEP(0) GL_STEP(1) GL_STEP(2) GL_STEP(3) GL_STEP(4) GL_STEP(5) GL_STEP(6) GL_STEP(7)
            GL_STEP(8) GL_STEP(9) GL_STEP(10) GL_STEP(11) GL_STEP(12) GL_STEP(13) GL_STEP(14) GL_STEP(15)
        }
    }
#undef GL_STEP
#undef GL_HALF
    // the last tile
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) store_piece(pc, c_live);
}

static bool gl_supported(int P, int Q, int Kc, int a_kmajor, int b_kmajor, int ab_dtype, int c_dtype) {
    if (!(ab_dtype == DM_BF16 || ab_dtype == DM_F16) || c_dtype != ab_dtype) return false;
    if (!a_kmajor || !b_kmajor) return false;
    if (P < 2048 || Q < GL_BN || Q % GL_BN || Kc < 512 || Kc % 512) return false;
    return true;
}

}  // namespace dm

extern "C" int dm_gemm_large_supported(int P, int Q, int Kc, int a_kmajor, int b_kmajor, int ab_dtype, int c_dtype) {
    return dm::gl_supported(P, Q, Kc, a_kmajor, b_kmajor, ab_dtype, c_dtype) ? 1 : 0;
}

extern "C" int dm_gemm_large(const dm_gemm_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gemm_large: null args"); return DM_ERR_ARG; }
    const dm_gemm_args& a = *args;
    if (!a.a || !a.b || !a.c) { set_error("dm_gemm_large: null tensor pointer"); return DM_ERR_ARG; }
    if (!gl_supported(a.P, a.Q, a.Kc, a.a_kmajor, a.b_kmajor, a.ab_dtype, a.c_dtype) || a.accumulate) {
        set_error("dm_gemm_large: unsupported (P %d Q %d Kc %d, kmajor %d %d, dtypes %d -> %d, accumulate %d): both operands k-major, "
                  "16-bit C of the operand dtype, P >= 2048, Q %% 256 == 0, Kc %% 512 == 0, no accumulation",
                  a.P, a.Q, a.Kc, a.a_kmajor, a.b_kmajor, a.ab_dtype, a.c_dtype, a.accumulate);
        return DM_ERR_ARG;
    }
    const int64_t amax = ((int64_t)(a.P - 1) * a.lda + a.Kc) * 2 + (int64_t)GL_BM * a.lda * 2, bmax = (int64_t)a.Q * a.ldb * 2,
                  cmax = ((int64_t)(a.P - 1) * a.ldc + a.Q) * 2 + (int64_t)GL_BM * a.ldc * 2;
    if (a.lda < a.Kc || a.ldb < a.Kc || a.ldc < a.Q || a.lda % 8 || a.ldb % 8 || a.ldc % 8 || ((uintptr_t)a.a % 16) || ((uintptr_t)a.b % 16) ||
        ((uintptr_t)a.c % 16) || amax >= (int64_t)0x7ffffff0 || bmax >= (int64_t)0x7ffffff0 || cmax >= (int64_t)0x7ffffff0) {
        set_error("dm_gemm_large: row strides must cover a row and be multiples of 8 elements, tensors 16-byte aligned and below 2 GB");
        return DM_ERR_LAYOUT;
    }
    // Whole rounds only: 256 persistent workgroups take 256 tiles at a time, so a tile count just above a multiple of 256 would cost a
    // whole extra round (out_proj forward: 392 row blocks x 2 column tiles = 3.06 rounds).  The row blocks beyond the last whole round
    // -- when they are less than half a round -- go to the small-launch kernel K11 (dm_gemm) behind this one, on the same stream.
    const int NT = a.Q / GL_BN, MB = (a.P + GL_BM - 1) / GL_BM;
    int q = 256, y = NT;
    while (y) { const int r = q % y; q = y; y = r; }     // gcd(256, NT)
    q = 256 / q;                                           // row blocks in a whole number of rounds
    int MB_main = MB / q * q;
    if (MB_main == 0 || (MB - MB_main) * NT >= 128) MB_main = MB;
    gl_args g;
    g.a = a.a; g.b = a.b; g.c = a.c;
    g.M = MB_main == MB ? a.P : MB_main * GL_BM;
    g.N = a.Q; g.K = a.Kc;
    g.lda = (int)a.lda; g.ldb = (int)a.ldb; g.ldc = (int)a.ldc;
    g.MB = MB_main;
    g.NT = NT;
    g.T = g.MB * g.NT;
    const int mb_x = (g.MB + 7) / 8;                       // whole row blocks per XCD: a row block's column tiles share one L2
    g.TX = mb_x * g.NT;
    hipStream_t st = (hipStream_t)stream;
    static const bool nt = [] { const char* e = getenv("DM_GL_NT"); return !e || atoi(e) != 0; }();      // DM_GL_NT=0: default-policy C stores (A/B runs)
    if (a.ab_dtype == DM_BF16 && !nt) hipLaunchKernelGGL((gemm_large_kernel<bf16_t, false>), dim3(256), dim3(GL_THREADS), 0, st, g);
    else if (a.ab_dtype == DM_BF16) hipLaunchKernelGGL((gemm_large_kernel<bf16_t>), dim3(256), dim3(GL_THREADS), 0, st, g);
    else hipLaunchKernelGGL((gemm_large_kernel<f16_t>), dim3(256), dim3(GL_THREADS), 0, st, g);
    if (MB_main != MB) {
        dm_gemm_args t = a;
        t.P = a.P - MB_main * GL_BM;
        t.a = (const char*)a.a + (int64_t)MB_main * GL_BM * a.lda * 2;
        t.c = (char*)a.c + (int64_t)MB_main * GL_BM * a.ldc * 2;
        const int rc = dm_gemm(&t, stream);
        if (rc != DM_OK) return rc;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gemm_large: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
