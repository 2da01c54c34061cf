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
__global__ __launch_bounds__(64 * DTP_WAVES) void dtproj_softplus_kernel(const dm_dtproj_args p) {
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

    const int tile0 = blockIdx.x * DTP_TILES;
    auto load_x = [&](int tile) -> dtp_u32x4 {
        int m = tile * 16 + j;
        m = (m < p.rows) ? m : p.rows - 1;
        dtp_u32x4 v = {0u, 0u, 0u, 0u};
        if (kvalid) v = *reinterpret_cast<const dtp_u32x4*>(X + (int64_t)m * p.xd_sr + 8 * g);
        return v;
    };
    dtp_u32x4 xf = load_x(tile0);
#pragma unroll 1
    for (int tt = 0; tt < DTP_TILES; ++tt) {
        const int tile = tile0 + tt;
        if (tile * 16 >= p.rows) break;                                        // wave-uniform
        const dtp_u32x4 xcur = xf;
        if (tt + 1 < DTP_TILES) xf = load_x(tile + 1);                          // clamped rows: always a legal address
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
    dim3 grid((tiles + DTP_TILES - 1) / DTP_TILES, (a.dim + DTP_WAVES * DTP_NQ * 64 - 1) / (DTP_WAVES * DTP_NQ * 64));
    if (grid.y > 65535) { set_error("dm_dtproj_softplus_fwd: dim too large"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    if (a.io_dtype == DM_BF16) hipLaunchKernelGGL((dtproj_softplus_kernel<bf16_t>), grid, dim3(64 * DTP_WAVES), 0, st, a);
    else hipLaunchKernelGGL((dtproj_softplus_kernel<f16_t>), grid, dim3(64 * DTP_WAVES), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_dtproj_softplus_fwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
