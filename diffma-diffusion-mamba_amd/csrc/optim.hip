// K14  dm_adamw_ema_step -- the optimiser step of the training loop, one pass over the parameters.
//
// Reference: train.py:153-166, 259-264 -- `opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0)`, `opt.step()`,
// `update_ema(ema, model.module)` (ema = decay * ema + (1 - decay) * param, train.py:36-47).  The reference trains at ONE sample per
// GPU (config/brain.yaml:11), where the whole step is ~8 ms and the parameter streaming -- torch's fused AdamW over 459 tensors
// (13 multi-tensor launches, 28 bytes per element), the EMA lerp (8 launches, 12 bytes) -- is a sixth of it.  Here every element
// is visited once: g, p, m, v, ema read; p, m, v, ema written (36 bytes), ONE launch for any number of tensors -- the tensor
// table lives in device memory, a workgroup looks up (tensor, chunk) by its index -- plus a one-thread-per-tensor launch that
// advances the step counters.  Arithmetic = torch's fused AdamW (aten/src/ATen/native/cuda/fused_adam_utils.cuh, ADAMW mode,
// amsgrad off, maximize off), fp32:
//     p -= lr wd p;   m = b1 m + (1 - b1) g;   v = b2 v + (1 - b2) g g;
//     p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps);          t = the tensor's step counter AFTER this step
//     ema += (1 - decay)(p - ema)                                           (torch._foreach_lerp_ for weights < 0.5)
// found_inf (device scalar, GradScaler's convention; the graphed step computes it from the all-reduced gradients): != 0 leaves p, m, v
// and the counters alone; the EMA still moves towards the unchanged weights when ema_on_skip is set (this repository's graphed
// step, graphed.py) and is left alone otherwise (the reference's eager loop `continue`s before update_ema, train.py:254-256).
#include "dm_common.h"

namespace dm {

constexpr int AW_THREADS = 256;
constexpr int AW_VPT = 4;                                   // 16-byte vectors per thread
constexpr int AW_CHUNK = AW_THREADS * AW_VPT * 4;           // elements per workgroup = DM_ADAMW_CHUNK

__global__ __launch_bounds__(64) void adamw_steps_kernel(const dm_adamw_args p) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= p.ntensors) return;
    if (p.found_inf && *p.found_inf != 0.0f) return;
    float* s = p.tensors[t].step;
    if (s) *s += 1.0f;
}

// torch's adam_math keeps lr, betas, weight decay and eps as DOUBLE scalars next to fp32 tensor elements: every expression below is
// evaluated in double and rounded to fp32 once on assignment, exactly as there (the kernel is HBM-bound; the fp64 pipe is idle anyway)
template <bool VEC>
__device__ __forceinline__ void adamw_elems(const dm_adamw_tensor& T, int64_t i0, int cnt, bool skip, bool ema_on, double lr, double wd, double b1, double b2,
                                            float bc1, float bc2_sqrt, double eps, float ema_w) {
    float g[4], w[4], m[4], v[4], e[4];
    if constexpr (VEC) {
        *reinterpret_cast<f32x4*>(w) = *reinterpret_cast<const f32x4*>(T.p + i0);
        if (T.ema && ema_on) *reinterpret_cast<f32x4*>(e) = *reinterpret_cast<const f32x4*>(T.ema + i0);
        if (!skip) {
            *reinterpret_cast<f32x4*>(g) = *reinterpret_cast<const f32x4*>(T.g + i0);
            *reinterpret_cast<f32x4*>(m) = *reinterpret_cast<const f32x4*>(T.m + i0);
            *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(T.v + i0);
        }
    } else {
        for (int j = 0; j < cnt; ++j) {
            w[j] = T.p[i0 + j];
            if (T.ema && ema_on) e[j] = T.ema[i0 + j];
            if (!skip) { g[j] = T.g[i0 + j]; m[j] = T.m[i0 + j]; v[j] = T.v[i0 + j]; }
        }
    }
    const float step_size = (float)(lr / (double)bc1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < cnt) {
            if (!skip) {
                if (wd != 0.0) w[j] = (float)((double)w[j] - lr * wd * (double)w[j]);
                m[j] = (float)(b1 * (double)m[j] + (1.0 - b1) * (double)g[j]);
                v[j] = (float)(b2 * (double)v[j] + (1.0 - b2) * (double)g[j] * (double)g[j]);
                const float denom = (float)((double)(__builtin_sqrtf(v[j]) / bc2_sqrt) + eps);
                w[j] -= step_size * m[j] / denom;
            }
            if (T.ema && ema_on) e[j] = (ema_w < 0.5f) ? e[j] + ema_w * (w[j] - e[j]) : w[j] - (w[j] - e[j]) * (1.0f - ema_w);      // at::lerp's two forms
        }
    }
    if constexpr (VEC) {
        if (!skip) {
            *reinterpret_cast<f32x4*>(T.p + i0) = *reinterpret_cast<const f32x4*>(w);
            *reinterpret_cast<f32x4*>(T.m + i0) = *reinterpret_cast<const f32x4*>(m);
            *reinterpret_cast<f32x4*>(T.v + i0) = *reinterpret_cast<const f32x4*>(v);
        }
        if (T.ema && ema_on) *reinterpret_cast<f32x4*>(T.ema + i0) = *reinterpret_cast<const f32x4*>(e);
    } else {
        for (int j = 0; j < cnt; ++j) {
            if (!skip) { T.p[i0 + j] = w[j]; T.m[i0 + j] = m[j]; T.v[i0 + j] = v[j]; }
            if (T.ema && ema_on) T.ema[i0 + j] = e[j];
        }
    }
}

__global__ __launch_bounds__(AW_THREADS) void adamw_ema_kernel(const dm_adamw_args p) {
    const int ti = p.block_tensor[blockIdx.x];
    const dm_adamw_tensor T = p.tensors[ti];
    const int64_t base = (int64_t)p.block_chunk[blockIdx.x] * AW_CHUNK;
    const bool skip = p.found_inf && *p.found_inf != 0.0f;
    const bool ema_on = !skip || p.ema_on_skip != 0;
    if (skip && !(T.ema && ema_on)) return;
    // the tensor's counter has been advanced by adamw_steps_kernel (same stream, earlier launch); the two bias corrections are
    // evaluated in double and rounded to fp32, as torch does
    const double t = T.step ? (double)*T.step : 1.0;
    const float bc1 = (float)(1.0 - pow(p.beta1, t));
    const float bc2 = (float)(1.0 - pow(p.beta2, t));
    const float bc2_sqrt = __builtin_sqrtf(bc2);
    const float ema_w = (float)(1.0 - p.ema_decay);
    const bool vec = ((((uintptr_t)T.p | (uintptr_t)T.m | (uintptr_t)T.v | (uintptr_t)T.g | (uintptr_t)T.ema) & 15) == 0);
#pragma unroll
    for (int k = 0; k < AW_VPT; ++k) {
        const int64_t i0 = base + ((int64_t)k * AW_THREADS + threadIdx.x) * 4;
        if (i0 >= T.n) break;
        const int cnt = (int)((T.n - i0) < 4 ? (T.n - i0) : 4);
        if (vec && cnt == 4) adamw_elems<true>(T, i0, 4, skip, ema_on, p.lr, p.weight_decay, p.beta1, p.beta2, bc1, bc2_sqrt, p.eps, ema_w);
        else adamw_elems<false>(T, i0, cnt, skip, ema_on, p.lr, p.weight_decay, p.beta1, p.beta2, bc1, bc2_sqrt, p.eps, ema_w);
    }
}

// found[0] = 1 if any gradient element of the table is Inf / NaN (the caller zeroes it first): what
// torch._amp_foreach_non_finite_check_and_unscale_ computes with 6 multi-tensor launches, as one
__global__ __launch_bounds__(AW_THREADS) void grads_nonfinite_kernel(const dm_adamw_args p) {
    const dm_adamw_tensor T = p.tensors[p.block_tensor[blockIdx.x]];
    const int64_t base = (int64_t)p.block_chunk[blockIdx.x] * AW_CHUNK;
    const bool vec = ((uintptr_t)T.g & 15) == 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < AW_VPT; ++k) {
        const int64_t i0 = base + ((int64_t)k * AW_THREADS + threadIdx.x) * 4;
        if (i0 >= T.n) break;
        const int cnt = (int)((T.n - i0) < 4 ? (T.n - i0) : 4);
        if (vec && cnt == 4) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(T.g + i0);
            // x - x is 0 for every finite x and NaN for Inf / NaN
            const float z = (g.x - g.x) + (g.y - g.y) + (g.z - g.z) + (g.w - g.w);
            bad = bad || !(z == 0.0f);
        } else {
            for (int j = 0; j < cnt; ++j) { const float x = T.g[i0 + j]; bad = bad || !((x - x) == 0.0f); }
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *p.nonfinite_out = 1.0f;
}

}  // namespace dm

extern "C" int dm_grads_nonfinite(const dm_adamw_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_grads_nonfinite: null args"); return DM_ERR_ARG; }
    const dm_adamw_args& a = *args;
    if (!a.tensors || !a.block_tensor || !a.block_chunk || !a.nonfinite_out) { set_error("dm_grads_nonfinite: null table / output pointer"); return DM_ERR_ARG; }
    if (a.ntensors <= 0 || a.nblocks <= 0) { set_error("dm_grads_nonfinite: non-positive ntensors / nblocks"); return DM_ERR_ARG; }
    hipLaunchKernelGGL(grads_nonfinite_kernel, dim3(a.nblocks), dim3(AW_THREADS), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_grads_nonfinite: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

extern "C" int dm_adamw_chunk(void) { return dm::AW_CHUNK; }

extern "C" int dm_adamw_ema_step(const dm_adamw_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_adamw_ema_step: null args"); return DM_ERR_ARG; }
    const dm_adamw_args& a = *args;
    if (!a.tensors || !a.block_tensor || !a.block_chunk) { set_error("dm_adamw_ema_step: null table pointer"); return DM_ERR_ARG; }
    if (a.ntensors <= 0 || a.nblocks <= 0) { set_error("dm_adamw_ema_step: non-positive ntensors / nblocks"); return DM_ERR_ARG; }
    if (!(a.beta1 >= 0.0 && a.beta1 < 1.0 && a.beta2 >= 0.0 && a.beta2 < 1.0) || !(a.eps >= 0.0) || !(a.ema_decay >= 0.0 && a.ema_decay <= 1.0)) {
        set_error("dm_adamw_ema_step: betas must lie in [0, 1), eps >= 0, ema_decay in [0, 1]"); return DM_ERR_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_steps_kernel, dim3((a.ntensors + 63) / 64), dim3(64), 0, st, a);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3(a.nblocks), dim3(AW_THREADS), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_adamw_ema_step: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
