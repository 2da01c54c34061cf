// K8  dm_dtproj_softplus_fwd -- delta = softplus(x_dbl[:, :R] @ dt_proj.weight^T + dt_proj.bias), gfx950.
//
// The dt_proj + softplus stage of mamba_inner_fn (reference call site block/mamba.py:346-348, parameters block/mamba.py:262-287;
// SURVEY.md A.1 steps 3-4).  Upstream the softplus is evaluated inside the selective scan, i.e. once per element in the
// forward AND once in the backward of every direction; here it rides in the epilogue of the rank-32 product that produces
// delta, whose cost is the WRITE of delta (rows x dim), and the VALU-bound scans run with DM_FLAG_DELTA_ACTIVATED.
//
// Mapping (wave64, v_mfma_f32_16x16x32_{bf16,f16}; K = 32 covers the whole reduction, so one instruction per 16 x 16 tile):
//   the operands are swapped -- A = 16 weight rows (output columns), B = 16 x_dbl rows -- so that accumulator register r of
//   lane l is column 4*(l>>4) + r of its tile for x_dbl row l&15.  The weight rows of the 4 tiles of a "quad" are picked so
//   that a lane ends up with 16 CONSECUTIVE output columns ( quad*64 + (l>>4)*16 + tile*4 + r ); the wave's 16 x 64 tile is
//   turned through a wave-private LDS slab and leaves as store instructions of 128 contiguous bytes per row (measured, MI355X,
//   1536 sequences: 179 us against 218 us for direct 16-byte stores from the accumulator layout and 184 us for 64-byte
//   pieces per row; two quads per wave: 192 us).  Weight fragments and the bias stay in registers for all row tiles of the
//   workgroup.  The write of delta (616 MB at that shape) is the floor: the library GEMM without the softplus takes 143-149 us.
#include "dm_common.h"
#include <cstdlib>
#include <type_traits>

namespace dm {

constexpr int DTP_WAVES = 4;      // waves per workgroup
constexpr int DTP_TILES = 8;      // row tiles (16 rows) per workgroup
constexpr int DTP_NQ = 1;         // quads (64 columns) per wave -> 256 columns per workgroup (grid.y covers dim)

typedef __bf16 dtp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 dtp_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t dtp_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct dtp_mfma;
template <> struct dtp_mfma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const dtp_u32x4& a, const dtp_u32x4& b) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dtp_bf16x8, a), __builtin_bit_cast(dtp_bf16x8, b), z, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return dm_cvt_pk_bf16(lo, hi); }
};
template <> struct dtp_mfma<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 run(const dtp_u32x4& a, const dtp_u32x4& b) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dtp_f16x8, a), __builtin_bit_cast(dtp_f16x8, b), z, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
};

// softplus for a 16-bit result: below e = 2^-12 log(1 + e) = e to 2^-13 relative, far inside the output rounding
__device__ __forceinline__ float softplus16_f(float x) {
    const float e = fast_exp2(x * LOG2E);
    const float big = fast_log2(1.0f + e) * LN2;
    const float r = (e < 2.44140625e-4f) ? e : big;
    return (x > 20.0f) ? x : r;
}

template <typename T>
__global__ __launch_bounds__(64 * DTP_WAVES) void dtproj_softplus_kernel(const mix_args<dm_dtproj_args> pm) {
    const dm_dtproj_args& p = pm.a[blockIdx.z];
    constexpr int LROW = DTP_NQ * 64 + 8;                                      // LDS row stride in elements (16 B of padding)
    __shared__ __attribute__((aligned(16))) T stage[DTP_WAVES][16][LROW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int cb = (blockIdx.y * DTP_WAVES + wave) * (DTP_NQ * 64);           // first column of this wave
    const T* __restrict__ W = (const T*)p.w;
    const T* __restrict__ X = (const T*)p.xdbl;
    T* __restrict__ O = (T*)p.delta;
    const int R = p.rank;
    const bool kvalid = 8 * g < R;                                             // rank % 8 == 0: a lane's 8 reduction elements are all in or all out

    // weight fragments: tile t of quad q, row i of the tile = output column cb + q*64 + (i>>2)*16 + t*4 + (i&3)
    auto colmap = [](int gg, int t, int r) { return gg * 16 + t * 4 + r; };
    dtp_u32x4 wf[DTP_NQ][4];
    float bias[DTP_NQ][16];
#pragma unroll
    for (int q = 0; q < DTP_NQ; ++q) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = cb + q * 64 + colmap(j >> 2, t, j & 3);
            wf[q][t] = (dtp_u32x4){0u, 0u, 0u, 0u};
            if (kvalid && col < p.dim) wf[q][t] = *reinterpret_cast<const dtp_u32x4*>(W + (int64_t)col * R + 8 * g);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int col = cb + q * 64 + colmap(g, e >> 2, e & 3);
            bias[q][e] = (p.bias && col < p.dim) ? p.bias[col] : 0.0f;
        }
    }

    // row tiles per workgroup: DTP_TILES for launches that fill the chip; a launch of few rows (the reference trains at ONE sample
    // per GPU: 37 tiles per mixer) is spread over more workgroups instead of walking 8 tiles one after the other in 20 of them
    const int tpw = (((p.rows + 15) >> 4) + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = blockIdx.x * tpw;
    auto load_x = [&](int tile) -> dtp_u32x4 {
        int m = tile * 16 + j;
        m = (m < p.rows) ? m : p.rows - 1;
        dtp_u32x4 v = {0u, 0u, 0u, 0u};
        if (kvalid) v = *reinterpret_cast<const dtp_u32x4*>(X + (int64_t)m * p.xd_sr + 8 * g);
        return v;
    };
    dtp_u32x4 xf = load_x(tile0);
#pragma unroll 1
    for (int tt = 0; tt < tpw; ++tt) {
        const int tile = tile0 + tt;
        if (tile * 16 >= p.rows) break;                                        // wave-uniform
        const dtp_u32x4 xcur = xf;
        if (tt + 1 < tpw) xf = load_x(tile + 1);                                // clamped rows: always a legal address
#pragma unroll
        for (int q = 0; q < DTP_NQ; ++q) {
            uint32_t pk[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 acc = dtp_mfma<T>::run(wf[q][t], xcur);
                const float v0 = softplus16_f(acc[0] + bias[q][4 * t + 0]);
                const float v1 = softplus16_f(acc[1] + bias[q][4 * t + 1]);
                const float v2 = softplus16_f(acc[2] + bias[q][4 * t + 2]);
                const float v3 = softplus16_f(acc[3] + bias[q][4 * t + 3]);
                pk[2 * t] = dtp_mfma<T>::pack(v0, v1);
                pk[2 * t + 1] = dtp_mfma<T>::pack(v2, v3);
            }
            {
                T* const srow = &stage[wave][j][q * 64 + g * 16];
                *reinterpret_cast<dtp_u32x4*>(srow) = (dtp_u32x4){pk[0], pk[1], pk[2], pk[3]};
                *reinterpret_cast<dtp_u32x4*>(srow + 8) = (dtp_u32x4){pk[4], pk[5], pk[6], pk[7]};
            }
        }
        {                                                                      // wave-private tile: no barrier, LDS ops of a wave are ordered
            constexpr int PIECES = DTP_NQ * 8;                                 // 16-B pieces per row of the wave's tile
            constexpr int ROWS_PER = 64 / PIECES;                              // rows per store instruction
#pragma unroll
            for (int s0 = 0; s0 < 16; s0 += ROWS_PER) {
                const int rr = s0 + lane / PIECES, pc = lane % PIECES;
                const dtp_u32x4 v = *reinterpret_cast<const dtp_u32x4*>(&stage[wave][rr][pc * 8]);
                const int mm = tile * 16 + rr, cc = cb + pc * 8;
                if (mm < p.rows && cc < p.dim) *reinterpret_cast<dtp_u32x4*>(O + (int64_t)mm * p.dim + cc) = v;
            }
        }
    }
}

}  // namespace dm

extern "C" int dm_dtproj_softplus_supported(int dim, int rank, int io_dtype) {
    return (io_dtype == DM_BF16 || io_dtype == DM_F16) && dim > 0 && dim % 16 == 0 && rank > 0 && rank <= 32 && rank % 8 == 0;
}

extern "C" int dm_dtproj_softplus_fwd(const dm_dtproj_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_dtproj_softplus_fwd: null args"); return DM_ERR_ARG; }
    const dm_dtproj_args& a = *args;
    if (!a.xdbl || !a.w || !a.delta) { set_error("dm_dtproj_softplus_fwd: null tensor pointer"); return DM_ERR_ARG; }
    if (a.rows <= 0 || a.dim <= 0 || a.rank <= 0) { set_error("dm_dtproj_softplus_fwd: non-positive size"); return DM_ERR_ARG; }
    if (!dm_dtproj_softplus_supported(a.dim, a.rank, a.io_dtype)) {
        set_error("dm_dtproj_softplus_fwd: needs 16-bit I/O, dim %% 16 == 0, rank %% 8 == 0, rank <= 32 (dim %d rank %d dtype %d)", a.dim, a.rank, a.io_dtype);
        return DM_ERR_LAYOUT;
    }
    if (a.xd_sr % 8 != 0 || a.xd_sr < a.rank || ((uintptr_t)a.xdbl % 16) || ((uintptr_t)a.w % 16) || ((uintptr_t)a.delta % 16)) {
        set_error("dm_dtproj_softplus_fwd: x_dbl row stride must be a multiple of 8 elements and all tensors 16-byte aligned"); return DM_ERR_LAYOUT;
    }
    const int tiles = (a.rows + 15) / 16;
    unsigned gz;
    const mix_args<dm_dtproj_args> m = mix_make(a, gz);
    dim3 grid((tiles + DTP_TILES - 1) / DTP_TILES, (a.dim + DTP_WAVES * DTP_NQ * 64 - 1) / (DTP_WAVES * DTP_NQ * 64), gz);
    for (int tpw = DTP_TILES / 2; tpw >= 1 && grid.x * grid.y * gz < 512; tpw /= 2) grid.x = (tiles + tpw - 1) / tpw;     // fewer tiles per workgroup until the chip is covered twice
    if (grid.y > 65535) { set_error("dm_dtproj_softplus_fwd: dim too large"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    if (a.io_dtype == DM_BF16) hipLaunchKernelGGL((dtproj_softplus_kernel<bf16_t>), grid, dim3(64 * DTP_WAVES), 0, st, m);
    else hipLaunchKernelGGL((dtproj_softplus_kernel<f16_t>), grid, dim3(64 * DTP_WAVES), 0, st, m);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_dtproj_softplus_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// K8b  dm_dtproj_bwd -- both consumers of d(delta) in ONE read of it (round 3).
//
// The backward of delta_raw = x_dbl[:, :R] @ W^T needs   dxdt[m][r] = sum_d dd[m][d] * W[d][r]   (reduction over the channels) and
// dW[d][r] = sum_m dd[m][d] * xdt[m][r]   (reduction over the rows).  As library GEMMs these are two passes over dd = d(delta_raw)
// ([3*B*L, d_inner]: 616 MB at the bench shape -- 129 us + 112 us, plus a strided copy of the first product into its x_dbl columns
// and the slab sum of the split-K second one).  Here a 512-thread workgroup walks 32-row tiles of dd; wave w owns the channels
// [w*CH, (w+1)*CH), CH = dim / 8:
//   * its 16-byte global loads of the tile ARE the MFMA fragments of the first product (K = channel is contiguous in memory);
//     operands swapped (A = W, B = dd) so that an accumulator register quad is 4 consecutive r of one row: 8-byte stores;
//     the 8 waves' partial [32 x R] tiles meet in LDS;
//   * the same registers go to a row-major LDS image of the tile, and `ds_read_b64_tr_b16` hands them back with K = row
//     (8 consecutive rows of one channel per lane): B fragments of  dW^T[r][d] += xdt^T[r][m] * dd[m][d]; the xdt tile takes the same
//     route.  The accumulators (2 x CH/16 tiles) stay in registers for all tiles of the workgroup and leave as ONE partial
//     [dim][R] fp32 image per workgroup (dm_colsum_f32 adds the images).
// The next tile's global loads are issued before the current tile's products.  Any row count (a ragged last tile reads row rows-1 for
// the missing rows and zeroes them at use), dim = 8 * CH with CH in {64, 96, 128},
// rank in {16, 32} (dm_dtproj_bwd_supported); everything else keeps the two GEMMs.
namespace dm {

constexpr int DTB_WAVES = 8, DTB_TM = 32;

template <typename T, int KC, int NR>      // KC = 32-channel chunks per wave (CH = 32 KC), NR = 16-wide r tiles (rank = 16 NR)
__global__ __launch_bounds__(64 * DTB_WAVES) void dtproj_bwd_kernel(const mix_args<dm_dtproj_bwd_args> pm) {
    const dm_dtproj_bwd_args& p = pm.a[blockIdx.z];
    constexpr int CH = 32 * KC, DIM = DTB_WAVES * CH, R = 16 * NR;
    constexpr int LROW = DIM + 16;                 // elements; 32 bytes of padding: the 4 rows x 32 bytes of a transpose read fall on distinct banks
    constexpr int XROW = R + 16;
    __shared__ __attribute__((aligned(16))) T tile[DTB_TM][LROW];
    __shared__ __attribute__((aligned(16))) T xt[DTB_TM][XROW];
    __shared__ __attribute__((aligned(16))) float osum[DTB_WAVES][2 * NR][64][4];      // [wave][m-tile of rows x r tile][lane][4 consecutive r]
    typedef short tr_v4s __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) tr_v4s* tr_ptr;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int d0 = wave * CH;
    const T* __restrict__ DD = (const T*)p.ddelta;
    const T* __restrict__ X = (const T*)p.xdbl;
    const T* __restrict__ W = (const T*)p.w;
    T* __restrict__ O = (T*)p.dxdbl;

    // W fragments (operand A of product 1): lane (j, g) of tile (kc, nr) holds W[d0 + 32 kc + 8 g .. + 8][16 nr + j]
    dtp_u32x4 wf[KC][NR];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const T* src = W + (int64_t)(d0 + 32 * kc + 8 * g + 2 * e) * R + 16 * nr + j;
                const uint32_t lo = __builtin_bit_cast(unsigned short, src[0]), hi = __builtin_bit_cast(unsigned short, src[R]);
                w[e] = lo | (hi << 16);
            }
            wf[kc][nr] = (dtp_u32x4){w[0], w[1], w[2], w[3]};
        }
    }
    f32x4 accw[NR][2 * KC];                        // dW^T tiles: rows = r (16 nr + 4 g + i), column = channel d0 + 16 nt + j
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int nt = 0; nt < 2 * KC; ++nt) accw[nr][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntile = (p.rows + DTB_TM - 1) / DTB_TM;             // the last tile may be ragged: its rows past the end read row rows-1 and are zeroed at use
    // fragments of a tile: [m-tile of 16 rows][kc]: lane (j, g) holds dd[tile*32 + 16 mt + j][d0 + 32 kc + 8 g .. + 8]
    auto load_tile = [&](int t, dtp_u32x4(&f)[2][KC], dtp_u32x4& xv) {
        t = (t < ntile) ? t : ntile - 1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                f[mt][kc] = *reinterpret_cast<const dtp_u32x4*>(DD + (int64_t)min(t * DTB_TM + 16 * mt + j, p.rows - 1) * DIM + d0 + 32 * kc + 8 * g);
        // the xdt tile [32][R] is R/8 16-byte pieces per row: the first 32 * R / 8 threads of the workgroup fetch one each
        const int pr = tid / (R / 8), pc = tid % (R / 8);
        xv = (dtp_u32x4){0u, 0u, 0u, 0u};
        if (pr < DTB_TM) xv = *reinterpret_cast<const dtp_u32x4*>(X + (int64_t)min(t * DTB_TM + pr, p.rows - 1) * p.xd_sr + 8 * pc);
    };
    dtp_u32x4 cur[2][KC], nxt[2][KC], xcur, xnxt;
    load_tile(blockIdx.x, cur, xcur);
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        load_tile(t + gridDim.x, nxt, xnxt);                       // lands while this tile is multiplied (clamped: always a legal tile)
        if (t == ntile - 1 && p.rows % DTB_TM != 0) {              // ragged last tile (wave-uniform): rows past the end contribute nothing
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                if (t * DTB_TM + 16 * mt + j >= p.rows) {
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc) cur[mt][kc] = (dtp_u32x4){0u, 0u, 0u, 0u};
                }
        }
        // ---- product 1, this wave's channel slice: acc[mt][nr] = W-fragment x dd-fragment, D[r][row]
        f32x4 acc1[2][NR];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) {
                    if constexpr (std::is_same<T, bf16_t>::value)
                        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dtp_bf16x8, wf[kc][nr]), __builtin_bit_cast(dtp_bf16x8, cur[mt][kc]), a, 0, 0, 0);
                    else
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dtp_f16x8, wf[kc][nr]), __builtin_bit_cast(dtp_f16x8, cur[mt][kc]), a, 0, 0, 0);
                }
                acc1[mt][nr] = a;
            }
        // ---- stage: the tile row-major, the xdt tile, this wave's partial of product 1
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) *reinterpret_cast<dtp_u32x4*>(&tile[16 * mt + j][d0 + 32 * kc + 8 * g]) = cur[mt][kc];
        {
            const int pr = tid / (R / 8), pc = tid % (R / 8);
            if (pr < DTB_TM) *reinterpret_cast<dtp_u32x4*>(&xt[pr][8 * pc]) = xcur;
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) *reinterpret_cast<f32x4*>(&osum[wave][mt * NR + nr][lane][0]) = acc1[mt][nr];
        __syncthreads();
        // ---- product 1: sum the 8 waves' partials; thread (tile q, lane l) owns rows r = 16 nr + 4 (l >> 4) + 0..3 of output row 16 mt + (l & 15)
        if (tid < 2 * NR * 64) {
            const int q = tid >> 6, l = tid & 63;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < DTB_WAVES; ++w) s += *reinterpret_cast<const f32x4*>(&osum[w][q][l][0]);
            const int mt = q / NR, nr = q % NR;
            const u32x2_t pk = {dtp_mfma<T>::pack(s[0], s[1]), dtp_mfma<T>::pack(s[2], s[3])};
            const int orow = t * DTB_TM + 16 * mt + (l & 15);
            if (orow < p.rows) *reinterpret_cast<u32x2_t*>(O + (int64_t)orow * p.dxd_sr + 16 * nr + 4 * (l >> 4)) = pk;
        }
        // ---- product 2: dW^T[r][d] += xdt^T[r][m] * dd[m][d], K = the tile's 32 rows; both operands through transposing LDS reads:
        //      lane (j, g) supplies the 8-byte piece (row 8 g + (j >> 2) [+ 4], columns c0 + 4 (j & 3) ..) and receives rows 8 g .. 8 g + 7 of column c0 + j
        dtp_u32x4 xa[NR];
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const tr_v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)&xt[8 * g + (j >> 2)][16 * nr + 4 * (j & 3)]);
            const tr_v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)&xt[8 * g + 4 + (j >> 2)][16 * nr + 4 * (j & 3)]);
            const u32x2_t w0 = __builtin_bit_cast(u32x2_t, a0), w1 = __builtin_bit_cast(u32x2_t, a1);
            xa[nr] = (dtp_u32x4){w0.x, w0.y, w1.x, w1.y};
        }
#pragma unroll
        for (int nt = 0; nt < 2 * KC; ++nt) {
            const tr_v4s b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)&tile[8 * g + (j >> 2)][d0 + 16 * nt + 4 * (j & 3)]);
            const tr_v4s b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_ptr)&tile[8 * g + 4 + (j >> 2)][d0 + 16 * nt + 4 * (j & 3)]);
            const u32x2_t w0 = __builtin_bit_cast(u32x2_t, b0), w1 = __builtin_bit_cast(u32x2_t, b1);
            const dtp_u32x4 bf = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) {
                if constexpr (std::is_same<T, bf16_t>::value)
                    accw[nr][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dtp_bf16x8, xa[nr]), __builtin_bit_cast(dtp_bf16x8, bf), accw[nr][nt], 0, 0, 0);
                else
                    accw[nr][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dtp_f16x8, xa[nr]), __builtin_bit_cast(dtp_f16x8, bf), accw[nr][nt], 0, 0, 0);
            }
        }
        __syncthreads();                                           // the images are rewritten by the next tile
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) cur[mt][kc] = nxt[mt][kc];
        xcur = xnxt;
    }
    // ---- the workgroup's dW image [dim][R]: register i of lane (j, g) of tile (nr, nt) is dW[d0 + 16 nt + j][16 nr + 4 g + i]
    float* part = p.part + (int64_t)blockIdx.x * DIM * R;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int nt = 0; nt < 2 * KC; ++nt) *reinterpret_cast<f32x4*>(part + (int64_t)(d0 + 16 * nt + j) * R + 16 * nr + 4 * g) = accw[nr][nt];
}

template <typename T>
static int dtproj_bwd_launch(const dm_dtproj_bwd_args& a, hipStream_t st) {
    unsigned gz;
    const mix_args<dm_dtproj_bwd_args> m = mix_make(a, gz);
    const dim3 grid((unsigned)a.nblk, 1, gz), block(64 * DTB_WAVES);
    const int kc = a.dim / (32 * DTB_WAVES), nr = a.rank / 16;
#define DM_DTB(KC, NR) hipLaunchKernelGGL((dtproj_bwd_kernel<T, KC, NR>), grid, block, 0, st, m)
    switch (kc * 10 + nr) {
        case 21: DM_DTB(2, 1); break;
        case 22: DM_DTB(2, 2); break;
        case 31: DM_DTB(3, 1); break;
        case 32: DM_DTB(3, 2); break;
        case 41: DM_DTB(4, 1); break;
        case 42: DM_DTB(4, 2); break;
        default: set_error("dm_dtproj_bwd: unsupported dim %d / rank %d", a.dim, a.rank); return DM_ERR_ARG;
    }
#undef DM_DTB
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_dtproj_bwd_supported(int dim, int rank, int io_dtype) {
    if (getenv("DIFFMA_DTPROJ_BWD_FUSED") && atoi(getenv("DIFFMA_DTPROJ_BWD_FUSED")) == 0) return 0;
    return (io_dtype == DM_BF16 || io_dtype == DM_F16) && (rank == 16 || rank == 32) && (dim == 512 || dim == 768 || dim == 1024);
}

extern "C" int dm_dtproj_bwd(const dm_dtproj_bwd_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_dtproj_bwd: null args"); return DM_ERR_ARG; }
    const dm_dtproj_bwd_args& a = *args;
    if (!a.ddelta || !a.xdbl || !a.w || !a.dxdbl || !a.part) { set_error("dm_dtproj_bwd: null tensor pointer"); return DM_ERR_ARG; }
    if (!dm_dtproj_bwd_supported(a.dim, a.rank, a.io_dtype)) { set_error("dm_dtproj_bwd: unsupported dim %d / rank %d / dtype %d", a.dim, a.rank, a.io_dtype); return DM_ERR_ARG; }
    if (a.rows <= 0) { set_error("dm_dtproj_bwd: rows (%d) must be positive", a.rows); return DM_ERR_ARG; }
    if (a.nblk <= 0 || a.nblk > (a.rows + DTB_TM - 1) / DTB_TM) { set_error("dm_dtproj_bwd: nblk (%d) must be in 1 .. ceil(rows / %d)", a.nblk, DTB_TM); return DM_ERR_ARG; }
    if (a.xd_sr % 8 != 0 || a.dxd_sr % 4 != 0 || ((uintptr_t)a.xdbl & 15) || ((uintptr_t)a.dxdbl & 7) || ((uintptr_t)a.ddelta & 15) || ((uintptr_t)a.part & 15)) {
        set_error("dm_dtproj_bwd: xdbl rows must be 16-byte aligned (stride % 8), dxdbl rows 8-byte aligned (stride % 4)");
        return DM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int rc = (a.io_dtype == DM_BF16) ? dtproj_bwd_launch<bf16_t>(a, st) : dtproj_bwd_launch<f16_t>(a, st);
    if (rc != DM_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_dtproj_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

// ---- n congruent launches in one (the two mixers of a block at small batch; see dm_common.h mix_args) ------------------------
extern "C" int dm_dtproj_softplus_fwd_n(const dm_dtproj_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_dtproj_softplus_fwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_dtproj_args* a) { return dm_dtproj_softplus_fwd(a, stream); },
                        [](const dm_dtproj_args& x, const dm_dtproj_args& y) {
                            return mix_congruent(x, y, &dm_dtproj_args::xdbl, &dm_dtproj_args::w, &dm_dtproj_args::bias, &dm_dtproj_args::delta);
                        });
}

extern "C" int dm_dtproj_bwd_n(const dm_dtproj_bwd_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_dtproj_bwd_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_dtproj_bwd_args* a) { return dm_dtproj_bwd(a, stream); },
                        [](const dm_dtproj_bwd_args& x, const dm_dtproj_bwd_args& y) {
                            return mix_congruent(x, y, &dm_dtproj_bwd_args::ddelta, &dm_dtproj_bwd_args::xdbl, &dm_dtproj_bwd_args::w,
                                                 &dm_dtproj_bwd_args::dxdbl, &dm_dtproj_bwd_args::part);
                        });
}
