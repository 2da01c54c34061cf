// Shared device helpers for the DiffMa gfx950 kernels.  CDNA4 only: wave64, no portability shims.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/diffma_hip.h"

namespace dm {

constexpr int WAVE = 64;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// ---- several congruent launches in one (dm_*_n entry points, include/diffma_hip.h) --------------------------------------
// The two mixers of a DiffMa block (reference block/mamba_block.py:107-108) run the same kernels on tensors of the same shape
// with different weights.  At the reference's own batch (config/brain.yaml: one sample per GPU) a training step is bound by
// the NUMBER of launches, so the small-launch kernels take an ARRAY of argument structs as their kernel argument and pick
// theirs with blockIdx.z (or a slice of it): one launch, grid.z = n, nothing else changes in the kernel bodies.
constexpr int DM_MAX_MIX = 2;
template <typename A> struct mix_args {
    A a[DM_MAX_MIX];
    int n;
};
// Host side: a dm_*_n entry point announces the second argument struct before it runs the ordinary single-launch path on the
// first; the launch site of a mix-capable kernel takes it (mix_make) -- any other launch site leaves it, and the entry point
// then launches the second struct on its own.  Thread-local: the backward runs on autograd's worker thread.
const void* mix_peek();
void mix_announce(const void* second);
bool mix_was_taken();
void mix_take();
template <typename A> static inline mix_args<A> mix_make(const A& a, unsigned& gz) {
    mix_args<A> m;
    m.a[0] = a;
    const A* second = static_cast<const A*>(mix_peek());
    if (second) {
        m.a[1] = *second;
        m.n = 2;
        mix_take();
    } else {
        m.a[1] = a;
        m.n = 1;
    }
    gz = (unsigned)m.n;
    return m;
}
// launch loop of a dm_*_n entry point: congruent neighbours share a launch when the kernel their shape selects can take two
template <typename A, typename F, typename C>
static inline int mix_launch_n(const A* args, int n, F single, C congruent) {
    int i = 0;
    while (i < n) {
        const bool pair = i + 1 < n && congruent(args[i], args[i + 1]);
        mix_announce(pair ? &args[i + 1] : nullptr);
        const int rc = single(&args[i]);
        const bool taken = mix_was_taken();
        mix_announce(nullptr);
        if (rc != 0) return rc;
        i += (pair && taken) ? 2 : 1;
    }
    return 0;
}
// all fields equal except the listed pointer members, whose null-ness must agree
template <typename A, typename... M>
static inline bool mix_congruent(const A& x, const A& y, M A::*... ptrs) {
    if ((... || ((x.*ptrs == nullptr) != (y.*ptrs == nullptr)))) return false;
    A u = x, v = y;
    ((u.*ptrs = nullptr, v.*ptrs = nullptr), ...);
    return __builtin_memcmp(&u, &v, sizeof(A)) == 0;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- storage types -------------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

// fp32 -> bf16 (round-to-nearest-even, NaN-safe) in ONE instruction: gfx950 has v_cvt_pk_bf16_f32 but no clang
// builtin for it; the software sequence is 7 VALU ops per store.
// v_cvt_pk_bf16_f32 emitted by the compiler (vector fptrunc): the scheduler knows its latency (K2 -2.5 %, K4x -5 %, K3x -3 %
// against the inline-asm form used before) and inserts the wait states a reader of MFMA results needs (an inline-asm reader
// gets none).  K1's SRD store and checkpoint pack keep the inline-asm form: that kernel measures 5 % SLOWER with this one.
typedef __bf16 dm_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float dm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t dm_cvt_pk_bf16(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((dm_f32x2_t){lo, hi}, dm_bf16x2_t));
}

__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x) {
    return dm_cvt_pk_bf16(x, x) & 0xffffu;
}

template <typename T> struct io;
template <> struct io<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float x) { *p = x; }
    // the element as loaded / its conversion at the point of use (a value fetched ahead must stay untouched until its consumer:
    // a conversion next to the load puts an s_waitcnt vmcnt(0) right behind it)
    static __device__ __forceinline__ uint32_t ld_raw(const float* p) { return __float_as_uint(*p); }
    static __device__ __forceinline__ float cv(uint32_t w) { return __uint_as_float(w); }
};
template <> struct io<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) {
        return __uint_as_float(((uint32_t)p->v) << 16);
    }
    static __device__ __forceinline__ void st(bf16_t* p, float x) { p->v = (uint16_t)f32_to_bf16_bits(x); }
    static __device__ __forceinline__ uint32_t ld_raw(const bf16_t* p) { return (uint32_t)p->v; }
    static __device__ __forceinline__ float cv(uint32_t w) { return __uint_as_float(w << 16); }
};
template <> struct io<f16_t> {
    static __device__ __forceinline__ float ld(const f16_t* p) { return (float)p->v; }
    static __device__ __forceinline__ void st(f16_t* p, float x) { p->v = (_Float16)x; }
    static __device__ __forceinline__ uint32_t ld_raw(const f16_t* p) { return (uint32_t)__builtin_bit_cast(unsigned short, p->v); }
    static __device__ __forceinline__ float cv(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)w); }
};

// Wave-uniform read-only operands (B_l, C_l rows, index tables) are read through the CONSTANT address
// space: a uniform constant-space load is selected as s_load_dword* (scalar cache -> SGPRs) instead of a
// 64-lane broadcast vector load.  Legal because nothing in a launch writes these buffers.
template <typename T> using cptr = const T __attribute__((address_space(4)))*;
template <typename T> __device__ __forceinline__ cptr<T> as_const(const T* p) {
    return (cptr<T>)(uintptr_t)p;
}
template <typename T> struct cio;
template <> struct cio<float> { static __device__ __forceinline__ float ld(cptr<float> p) { return *p; } };
template <> struct cio<bf16_t> {
    static __device__ __forceinline__ float ld(cptr<bf16_t> p) { return __uint_as_float(((uint32_t)p->v) << 16); }
};
template <> struct cio<f16_t> { static __device__ __forceinline__ float ld(cptr<f16_t> p) { return (float)p->v; } };

// ---- buffer (SRD) addressing ---------------------------------------------------------------------
// address = base(SGPRx4) + voffset(VGPR, bytes) + soffset(SGPR, bytes): the per-lane part (channel * size)
// is ONE register shared by every tensor, and the wave-uniform row offset rides in an SGPR, so a row
// access costs one s_add/s_mul -- no 64-bit scalar multiply, no v_lshl_add_u64 per access.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, /*num_records*/ -1, 0x00020000);
}
// 2 GB window: a lane whose VGPR offset is BIO_OOB is out of range -- its load returns 0 and its store is dropped.  Lets a kernel
// predicate a memory access per lane (or per wave) WITHOUT a branch around it: hipcc's s_waitcnt bookkeeping loses count at every
// control-flow join whose arms issued different numbers of memory operations, and a prefetch pipeline depends on exact counts.
constexpr int BIO_OOB = (int)0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc_2g(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, /*num_records*/ (int)0x80000000u, 0x00020000);
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// W 32-bit words of a lane (W % 4 == 0) as 16-byte buffer accesses, group g of 4 words at byte offset g * gstride: with
// gstride = 16 * (lanes in a row) every instruction reads / writes one dense run of 16-byte pieces.  (The lane-contiguous form,
// 32 bytes per lane written by two instructions at a 32-byte lane stride, halves the instruction count as well but leaves every
// 32-byte sector half-written per instruction: the forward scan got 7.5 % slower with it.)
#ifndef DM_CK_ST_AUX
#define DM_CK_ST_AUX 0         // cache-policy bits of the checkpoint stores / loads (developer A/B: 2 = nt, 1 = sc0, 16 = sc1)
#endif
#ifndef DM_CK_LD_AUX
#define DM_CK_LD_AUX 0
#endif
template <int W>
__device__ __forceinline__ void bio_st_words(const uint32_t (&w)[W], rsrc_t r, int voff, int soff, int gstride) {
    static_assert(W % 4 == 0, "whole 16-byte groups");
#pragma unroll
    for (int g = 0; g < W / 4; ++g) {
        const u32x4_t q = {w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff + g * gstride, DM_CK_ST_AUX);
    }
}
template <int W>
__device__ __forceinline__ void bio_ld_words(uint32_t (&w)[W], rsrc_t r, int voff, int soff, int gstride) {
    static_assert(W % 4 == 0, "whole 16-byte groups");
#pragma unroll
    for (int g = 0; g < W / 4; ++g) {
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff + g * gstride, DM_CK_LD_AUX);
        w[4 * g] = q[0]; w[4 * g + 1] = q[1]; w[4 * g + 2] = q[2]; w[4 * g + 3] = q[3];
    }
}
#ifndef DM_BIO_LD_AUX
#define DM_BIO_LD_AUX 0        // cache-policy bits of the element loads / stores through bio<T> in this translation unit (2 = nt: streamed once)
#endif
#ifndef DM_BIO_ST_AUX
#define DM_BIO_ST_AUX 0
#endif
template <typename T> struct bio;
template <> struct bio<float> {
    static __device__ __forceinline__ float ld(rsrc_t r, int voff, int soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, DM_BIO_LD_AUX));
    }
    // the loaded word as it is (a prefetched value must not be touched before its consumer: any arithmetic on it puts the
    // wait for the load right behind the load) and its conversion at the point of use
    typedef uint32_t raw_t;
    static __device__ __forceinline__ raw_t ld_raw(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, DM_BIO_LD_AUX); }
    static __device__ __forceinline__ float cv(raw_t w) { return __uint_as_float(w); }
    static __device__ __forceinline__ void st(rsrc_t r, int voff, int soff, float x) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, DM_BIO_ST_AUX);
    }
    static __device__ __forceinline__ void st_cv(rsrc_t r, int voff, int soff, float x) { st(r, voff, soff, x); }
};
template <> struct bio<bf16_t> {
    static __device__ __forceinline__ float ld(rsrc_t r, int voff, int soff) {
        return __uint_as_float(((uint32_t)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, DM_BIO_LD_AUX)) << 16);
    }
    typedef unsigned short raw_t;
    static __device__ __forceinline__ raw_t ld_raw(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, DM_BIO_LD_AUX); }
    static __device__ __forceinline__ float cv(raw_t w) { return __uint_as_float((uint32_t)w << 16); }
    static __device__ __forceinline__ void st(rsrc_t r, int voff, int soff, float x) {
        uint32_t b;
        asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(b) : "v"(x));       // low half is what store_b16 writes
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)b, r, voff, soff, DM_BIO_ST_AUX);
    }
    static __device__ __forceinline__ void st_cv(rsrc_t r, int voff, int soff, float x) {      // conversion visible to the scheduler
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)dm_cvt_pk_bf16(x, x), r, voff, soff, DM_BIO_ST_AUX);
    }
};
template <> struct bio<f16_t> {
    static __device__ __forceinline__ float ld(rsrc_t r, int voff, int soff) {
        const unsigned short b = __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, DM_BIO_LD_AUX);
        _Float16 h;
        __builtin_memcpy(&h, &b, 2);
        return (float)h;
    }
    typedef unsigned short raw_t;
    static __device__ __forceinline__ raw_t ld_raw(rsrc_t r, int voff, int soff) { return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, DM_BIO_LD_AUX); }
    static __device__ __forceinline__ float cv(raw_t w) {
        const unsigned short b = w;
        _Float16 h;
        __builtin_memcpy(&h, &b, 2);
        return (float)h;
    }
    static __device__ __forceinline__ void st(rsrc_t r, int voff, int soff, float x) {
        const _Float16 h = (_Float16)x;
        unsigned short b;
        __builtin_memcpy(&b, &h, 2);
        __builtin_amdgcn_raw_buffer_store_b16(b, r, voff, soff, DM_BIO_ST_AUX);
    }
    static __device__ __forceinline__ void st_cv(rsrc_t r, int voff, int soff, float x) { st(r, voff, soff, x); }
};
// NS consecutive elements (NS*sizeof(T) in {4, 8, 16} bytes use one wide load)
template <typename T, int NS>
__device__ __forceinline__ void bio_ld_vec(float (&v)[NS], rsrc_t r, int voff, int soff) {
    constexpr int BYTES = NS * (int)sizeof(T);
    if constexpr (BYTES == 16) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        alignas(16) uint32_t w[4] = {q[0], q[1], q[2], q[3]};
        const T* e = reinterpret_cast<const T*>(w);
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = io<T>::ld(e + k);
    } else if constexpr (BYTES == 8) {
        const auto q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        alignas(8) uint32_t w[2] = {q[0], q[1]};
        const T* e = reinterpret_cast<const T*>(w);
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = io<T>::ld(e + k);
    } else if constexpr (BYTES == 32) {
        float lo[NS / 2], hi[NS / 2];
        bio_ld_vec<T, NS / 2>(lo, r, voff, soff);
        bio_ld_vec<T, NS / 2>(hi, r, voff + 16, soff);
#pragma unroll
        for (int k = 0; k < NS / 2; ++k) { v[k] = lo[k]; v[NS / 2 + k] = hi[k]; }
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) v[k] = bio<T>::ld(r, voff + k * (int)sizeof(T), soff);
    }
}

// ---- transcendental helpers (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp, no range fix-ups) ----
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// softplus(x) = log(1+e^x); identity above 20 like the reference operator.
__device__ __forceinline__ float softplus_f(float x) {
    float e = fast_exp2(x * LOG2E);
    float big = fast_log2(1.0f + e) * LN2;
    float small = e * (1.0f - 0.5f * e);          // log1p series, exact to fp32 for e < 2^-12
    float r = (e < 2.44140625e-4f) ? small : big;
    return (x > 20.0f) ? x : r;
}
__device__ __forceinline__ float sigmoid_f(float x) {
    return fast_rcp(1.0f + fast_exp2(-x * LOG2E));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

// thread-local error string (host side)
void set_error(const char* fmt, ...);

// Sum over the 64 lanes of a wave on the VALU's DPP path (every lane gets the total): 4 in-row butterfly steps, two
// row broadcasts and one readlane, instead of 6 ds_bpermute round trips through the LDS pipe (__shfl_xor).
__device__ __forceinline__ float wave_sum_dpp(float x) {
#define DM_DPP_ADD(CTRL, ROWMASK) x += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, ROWMASK, 0xF, true))
    DM_DPP_ADD(0xB1, 0xF);        // quad_perm [1,0,3,2]   : lane ^ 1
    DM_DPP_ADD(0x4E, 0xF);        // quad_perm [2,3,0,1]   : lane ^ 2
    DM_DPP_ADD(0x141, 0xF);       // row_half_mirror       : the other quad of the 8
    DM_DPP_ADD(0x140, 0xF);       // row_mirror            : the other half of the 16-lane row -> every lane = row total
    DM_DPP_ADD(0x142, 0xA);       // row_bcast15 into rows 1, 3 : += total of the row below
    DM_DPP_ADD(0x143, 0xC);       // row_bcast31 into rows 2, 3 : += total of rows 0..1
#undef DM_DPP_ADD
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(x), 63));
}

}  // namespace dm
