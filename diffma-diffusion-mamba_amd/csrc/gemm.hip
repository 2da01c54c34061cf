// K11  dm_gemm / dm_gemm_n -- C[P][Q] = opA(A)[P][Kc] * opB(B)[Kc][Q], 16-bit operands, fp32 accumulation, gfx950.
//
// The dense projections of the Mamba mixer (reference block/mamba.py:261 in_proj, :315 out_proj, called at :333-337 and inside
// mamba_inner_fn) and their two gradients, for the SMALL-LAUNCH regime: the reference's own configuration trains at one sample per
// GPU (config/brain.yaml:11), i.e. M = 196 rows, where
//   * a training step is made of launches, and the two mixers of a block (block/mamba_block.py:107-108) want each product ONCE for
//     both -- the kernel takes an ARRAY of argument structs and picks its own by blockIdx.z (dm_gemm_n, dm_common.h mix_args);
//   * the vendor library cannot be trusted with the batch-2 form: torch.bmm of [2, 12544, 1024] x [2, 1024, 512] returns NaN with
//     the library's default kernel, and TunableOp's tuning loop faults on three more of these shapes (MI355X, ROCm 7.2; probe:
//     tools/dbg_bmm.py);
//   * its single GEMMs take 12-20 us each at M = 1568 (tile quantisation: a 256 x 256 tile grid does not fill 256 CUs).
// Large batches stay on the library (solution table, split-K): this kernel is a 2-barrier LDS-staged structure, not a
// 256 x 256 8-phase pipeline.
//
// One kernel for the three products of a Linear layer -- what differs is how an operand lies in memory:
//   k-major   : stored [rows][Kc], the contraction index contiguous  (x and W in the forward y = x W^T)
//   row-major : stored [Kc][rows], the OTHER index contiguous        (W in dx = dy W; dy and x in dW = dy^T x)
// Both end up in LDS as [row][32 k] images (80-byte rows: 16 bytes of padding keep the 16-lane fragment reads off each other's
// banks); a k-major tile is copied with 16-byte accesses, a row-major tile is loaded with 16-byte accesses along its contiguous
// index, stored as it is, and transposed by `ds_read_b64_tr_b16` fragment reads.  Fragments for v_mfma_f32_16x16x32_{bf16,f16}: lane (i = l & 15, g = l >> 4)
// reads the 16 bytes k = 8g .. 8g+7 of row i -- the same mapping for both operands, so the product is issued as B-rows x A-rows
// (operands swapped): accumulator register r of lane l is then C[m = l & 15][n = 4 (l >> 4) + r], four CONSECUTIVE columns of
// one output row, and leaves as one 8-byte (16-bit C) or 16-byte (fp32 C) store.
// Workgroup = 4 waves (2 x 2) on a BM x BN tile, BK = 32; the next tile's global loads are in flight (registers) while the
// current one is multiplied.
#include "dm_common.h"
#include <cstdlib>
#include <type_traits>

namespace dm {

// LDS image of an operand tile: [row][BK k] with 16 bytes of padding per row; the 16-byte k-pieces of a row are XOR-swizzled with
// bits 3.. of the row, so that the transposing 2-byte stores of a row-major tile (neighbouring lanes 8 rows apart at one k: the same
// bank unswizzled) spread over the banks.  BK = 128 for small grids: a K-step then carries 4x the work per memory latency -- with one
// tile of prefetch a 64 x 64 x 32 step was ~1.3 us, i.e. the kernel waited for its loads (first version: 20.8 us for both mixers'
// in_proj at 1568 rows).
template <int BK> struct gm_lds {
    static constexpr int LROW = BK + 8;
    static __device__ __forceinline__ int off(int row, int k) { return row * LROW + ((((k >> 3) ^ (row >> 3)) & (BK / 8 - 1)) << 3) + (k & 7); }
};

// A row-major operand tile stays row-major in LDS ([BK contraction rows][ROWS + 16]: 16-byte stores as loaded) and its fragments
// come back through `ds_read_b64_tr_b16`, which hands lane (j = l & 15, g = l >> 4) the 4 x 4 transposed block it needs: two reads
// give rows 8g .. 8g+7 (the contraction) of column c0 + j.  (First version: 2-byte transposing stores, 8 per piece -- the weight
// gradient, whose two operands are both row-major, took 29 us for both mixers' in_proj at 1568 rows.)
template <int ROWS> struct gm_rm {
    static constexpr int LROW = ROWS + 16;          // 32 bytes of padding: the 4 rows x 32 bytes of a transpose read fall on distinct banks
};
typedef short gm_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) gm_v4s* gm_tr_ptr;
typedef uint32_t gm_u32x2 __attribute__((ext_vector_type(2)));

typedef __bf16 gm_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gm_f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t gm_u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct gm_mfma;
template <> struct gm_mfma<bf16_t> {
    static __device__ __forceinline__ f32x4 run(const gm_u32x4& a, const gm_u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gm_bf16x8, a), __builtin_bit_cast(gm_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) { return dm_cvt_pk_bf16(lo, hi); }
};
template <> struct gm_mfma<f16_t> {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 run(const gm_u32x4& a, const gm_u32x4& b, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gm_f16x8, a), __builtin_bit_cast(gm_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ uint32_t pack(float lo, float hi) {
        h2 v;
        v.x = (_Float16)lo;
        v.y = (_Float16)hi;
        return __builtin_bit_cast(uint32_t, v);
    }
};

// One operand tile: ROWS rows x BK k.  KMAJOR: 16-byte pieces along k (ROWS * BK / 8 pieces); row-major source: 16-byte pieces along
// the row index at one k (BK * ROWS / 8 pieces).  256 threads -> ROWS * BK / 2048 pieces per thread either way.
template <typename T, int ROWS, int BK, bool KMAJOR>
struct gm_tile {
    static constexpr int NP = ROWS * BK / 2048, KP = BK / 8;
    gm_u32x4 v[NP];
    static __device__ __forceinline__ void where(int q, int& row, int& k) {
        if (KMAJOR) { row = q / KP; k = (q % KP) * 8; }                              // KP pieces of 8 k per row
        else { k = (q >> 2) & (BK - 1); row = ((q / (4 * BK)) * 4 + (q & 3)) * 8; }  // 4 neighbouring lanes read 64 contiguous bytes of one k
    }
    // rows [r0, r0 + ROWS) of the operand (its own row count `nrows`), k in [k0, k0 + BK) of Kc; ld = row stride of the STORED matrix
    __device__ __forceinline__ void load(const T* __restrict__ base, int64_t ld, int r0, int nrows, int k0, int Kc, int tid) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int row, k;
            where(tid + p * 256, row, k);
            const bool ok = (r0 + row < nrows) && (k0 + k < Kc);          // (sizes are multiples of 8 along the contiguous index)
            const int64_t off = KMAJOR ? (int64_t)(r0 + row) * ld + (k0 + k) : (int64_t)(k0 + k) * ld + (r0 + row);
            const gm_u32x4 z = {0u, 0u, 0u, 0u};
            v[p] = ok ? *reinterpret_cast<const gm_u32x4*>(base + off) : z;
        }
    }
    __device__ __forceinline__ void store(uint16_t* lds, int tid) const {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            int row, k;
            where(tid + p * 256, row, k);
            if (KMAJOR) *reinterpret_cast<gm_u32x4*>(lds + gm_lds<BK>::off(row, k)) = v[p];
            else *reinterpret_cast<gm_u32x4*>(lds + k * gm_rm<ROWS>::LROW + row) = v[p];          // as loaded: [k][row .. row + 7]
        }
    }
};

// fragment of operand rows r0 + (l & 15), contraction kk + 8 (l >> 4) .. + 7, from either LDS image
template <int ROWS, int BK, bool KMAJOR>
__device__ __forceinline__ gm_u32x4 gm_frag(const uint16_t* lds, int r0, int kk, int fi, int fg) {
    if constexpr (KMAJOR) {
        return *reinterpret_cast<const gm_u32x4*>(lds + gm_lds<BK>::off(r0 + fi, kk + 8 * fg));
    } else {
        const uint16_t* q = lds + (kk + 8 * fg + (fi >> 2)) * gm_rm<ROWS>::LROW + r0 + 4 * (fi & 3);
        const gm_v4s a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_tr_ptr)q);
        const gm_v4s a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((gm_tr_ptr)(q + 4 * gm_rm<ROWS>::LROW));
        const gm_u32x2 w0 = __builtin_bit_cast(gm_u32x2, a0), w1 = __builtin_bit_cast(gm_u32x2, a1);
        return (gm_u32x4){w0.x, w0.y, w1.x, w1.y};
    }
}

template <typename T, typename TC, int BM, int BN, int BK, bool AK, bool BKM>
__global__ __launch_bounds__(256) void gemm_kernel(const mix_args<dm_gemm_args> pm) {
    const dm_gemm_args& p = pm.a[blockIdx.z];
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;       // wave tile and its 16 x 16 sub-tiles
    __shared__ __attribute__((aligned(16))) uint16_t As[AK ? BM * gm_lds<BK>::LROW : BK * gm_rm<BM>::LROW];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[BKM ? BN * gm_lds<BK>::LROW : BK * gm_rm<BN>::LROW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int P = p.P, Q = p.Q, Kc = p.Kc;
    const T* __restrict__ A = (const T*)p.a;
    const T* __restrict__ B = (const T*)p.b;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    gm_tile<T, BM, BK, AK> ta;
    gm_tile<T, BN, BK, BKM> tb;
    ta.load(A, p.lda, m0, P, 0, Kc, tid);
    tb.load(B, p.ldb, n0, Q, 0, Kc, tid);
    const int fi = lane & 15, fg = lane >> 4;
    for (int k0 = 0; k0 < Kc; k0 += BK) {
        __syncthreads();                                   // the previous step's fragment reads are done
        ta.store(As, tid);
        tb.store(Bs, tid);
        __syncthreads();
        if (k0 + BK < Kc) {                                // next tile: in flight during the products
            ta.load(A, p.lda, m0, P, k0 + BK, Kc, tid);
            tb.load(B, p.ldb, n0, Q, k0 + BK, Kc, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            gm_u32x4 fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TM; ++j) fa[j] = gm_frag<BM, BK, AK>(As, wm + 16 * j, kk, fi, fg);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[i] = gm_frag<BN, BK, BKM>(Bs, wn + 16 * i, kk, fi, fg);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = gm_mfma<T>::run(fb[i], fa[j], acc[i][j]);      // rows = n, columns = m
        }
    }
    // lane l holds C[m = .. + (l & 15)][n = .. + 4 (l >> 4) + r], r = 0..3
    TC* __restrict__ C = (TC*)p.c;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm + 16 * j + fi;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int n = n0 + wn + 16 * i + 4 * fg;
            if (m < P && n < Q) {                          // Q % 4 == 0
                TC* dst = C + (int64_t)m * p.ldc + n;
                f32x4 v = acc[i][j];
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                if (p.accumulate) {                        // wave-uniform
                    if constexpr (std::is_same<TC, float>::value) {
                        v += *reinterpret_cast<const f32x4*>(dst);
                    } else {
                        const u32x2 o = *reinterpret_cast<const u32x2*>(dst);
                        const TC* e = reinterpret_cast<const TC*>(&o);
                        v.x += io<TC>::ld(e); v.y += io<TC>::ld(e + 1); v.z += io<TC>::ld(e + 2); v.w += io<TC>::ld(e + 3);
                    }
                }
                if constexpr (std::is_same<TC, float>::value) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {
                    const u32x2 w = {gm_mfma<T>::pack(v.x, v.y), gm_mfma<T>::pack(v.z, v.w)};
                    *reinterpret_cast<u32x2*>(dst) = w;
                }
            }
        }
    }
}

template <typename T, typename TC, int BM, int BN, int BK>
static void gemm_launch_l(const dm_gemm_args& a, hipStream_t st) {
    unsigned gz;
    const mix_args<dm_gemm_args> m = mix_make(a, gz);
    dim3 grid((a.Q + BN - 1) / BN, (a.P + BM - 1) / BM, gz), block(256);
    if (a.a_kmajor) {
        if (a.b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, BM, BN, BK, true, true>), grid, block, 0, st, m);
        else hipLaunchKernelGGL((gemm_kernel<T, TC, BM, BN, BK, true, false>), grid, block, 0, st, m);
    } else {
        if (a.b_kmajor) hipLaunchKernelGGL((gemm_kernel<T, TC, BM, BN, BK, false, true>), grid, block, 0, st, m);
        else hipLaunchKernelGGL((gemm_kernel<T, TC, BM, BN, BK, false, false>), grid, block, 0, st, m);
    }
}

template <typename T, typename TC>
static void gemm_launch_t(const dm_gemm_args& a, hipStream_t st) {
    // Tile by the size of the grid (both mixers counted when a second struct is announced): the largest tile that still gives every CU
    // a workgroup -- these launches are bound by the L2 -> CU traffic of their operands (a 64 x 64 tile uses a loaded byte for 64
    // multiply-adds, a 128 x 128 tile for 128), not by the matrix pipe.  64 x 64 (BK 128) is the fallback for the smallest grids.
    static const int force = [] { const char* e = getenv("DM_GEMM_TILE"); return e ? atoi(e) : 0; }();      // developer override: 1 / 2 / 3
    const int64_t n = mix_peek() ? 2 : 1;
    auto wgs = [&](int bm, int bn) { return n * (int64_t)((a.P + bm - 1) / bm) * ((a.Q + bn - 1) / bn); };
    int pick = force;
    if (!pick) pick = wgs(128, 128) >= 256 ? 3 : (wgs(128, 64) >= 256 ? 2 : 1);
    if (pick == 3) gemm_launch_l<T, TC, 128, 128, 64>(a, st);
    else if (pick == 2) gemm_launch_l<T, TC, 128, 64, 64>(a, st);
    else gemm_launch_l<T, TC, 64, 64, 128>(a, st);
}

}  // namespace dm

extern "C" int dm_gemm_supported(int P, int Q, int Kc, int a_kmajor, int b_kmajor, int ab_dtype, int c_dtype) {
    if (!(ab_dtype == DM_BF16 || ab_dtype == DM_F16) || !(c_dtype == DM_F32 || c_dtype == ab_dtype)) return 0;
    if (P <= 0 || Q <= 0 || Kc <= 0 || Q % 8 != 0) return 0;
    if ((a_kmajor || b_kmajor) && Kc % 8 != 0) return 0;        // 16-byte pieces along the contraction
    if (!a_kmajor && P % 8 != 0) return 0;                       // ... or along the operand's own rows
    return 1;
}

extern "C" int dm_gemm(const dm_gemm_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_gemm: null args"); return DM_ERR_ARG; }
    const dm_gemm_args& a = *args;
    if (!a.a || !a.b || !a.c) { set_error("dm_gemm: null tensor pointer"); return DM_ERR_ARG; }
    if (!dm_gemm_supported(a.P, a.Q, a.Kc, a.a_kmajor, a.b_kmajor, a.ab_dtype, a.c_dtype)) {
        set_error("dm_gemm: unsupported shape / dtype (P %d Q %d Kc %d, a_kmajor %d b_kmajor %d, dtypes %d -> %d): 16-bit operands, "
                  "Q %% 8 == 0, and the contiguous index of every operand a multiple of 8", a.P, a.Q, a.Kc, a.a_kmajor, a.b_kmajor, a.ab_dtype, a.c_dtype);
        return DM_ERR_ARG;
    }
    const int64_t a_min = a.a_kmajor ? a.Kc : a.P, b_min = a.b_kmajor ? a.Kc : a.Q;
    if (a.lda < a_min || a.ldb < b_min || a.ldc < a.Q || a.lda % 8 || a.ldb % 8 || a.ldc % 4 ||
        ((uintptr_t)a.a % 16) || ((uintptr_t)a.b % 16) || ((uintptr_t)a.c % 16)) {
        set_error("dm_gemm: row strides must cover a row, be multiples of 8 (operands) / 4 (C) elements, and the tensors 16-byte aligned");
        return DM_ERR_LAYOUT;
    }
    if ((a.P + 63) / 64 > 65535) { set_error("dm_gemm: P too large"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const bool c32 = a.c_dtype == DM_F32;
    if (a.ab_dtype == DM_BF16) { if (c32) gemm_launch_t<bf16_t, float>(a, st); else gemm_launch_t<bf16_t, bf16_t>(a, st); }
    else { if (c32) gemm_launch_t<f16_t, float>(a, st); else gemm_launch_t<f16_t, f16_t>(a, st); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_gemm: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

extern "C" int dm_gemm_n(const dm_gemm_args* args, int n, void* stream) {
    using namespace dm;
    if (!args || n <= 0) { set_error("dm_gemm_n: null args / n <= 0"); return DM_ERR_ARG; }
    return mix_launch_n(args, n, [&](const dm_gemm_args* a) { return dm_gemm(a, stream); },
                        [](const dm_gemm_args& x, const dm_gemm_args& y) {
                            // dm_gemm validates args[i] only: a second struct rides along only with the same 16-byte alignment
                            auto al = [](const void* p, const void* q) { return (((uintptr_t)p ^ (uintptr_t)q) & 15) == 0; };
                            return al(x.a, y.a) && al(x.b, y.b) && al(x.c, y.c) &&
                                   mix_congruent(x, y, &dm_gemm_args::a, &dm_gemm_args::b, &dm_gemm_args::c);
                        });
}
