// K15  dm_q_sample / dm_training_loss -- the diffusion wrapper around the denoiser call of a TRAINING step, two launches.
//
// Reference: GaussianDiffusion.training_losses (diffusion/gaussian_diffusion.py:715-789) for the configuration DiffMa trains with --
// create_diffusion("") = epsilon prediction, learned-range variance, MSE loss with the variational bound training the variance only
// (the mean prediction enters the bound detached, :752-766):
//     x_t   = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise                                   (q_sample, :215-230)
//     eps, v = model(x_t, t)[:, :C], [:, C:]
//     mse_b = mean_i (noise - eps)^2
//     logvar = frac log(beta_t) + (1 - frac) log(posterior_var_t),  frac = (v + 1) / 2      (p_mean_variance, :254-330)
//     mean  = coef1_t (sqrt(1/abar_t) x_t - sqrt(1/abar_t - 1) eps) + coef2_t x_t,   true_mean = coef1_t x_0 + coef2_t x_t
//     vb_b  = mean_i [ t > 0 ? KL(N(true_mean, posterior_var_t) || N(mean, e^logvar))           (normal_kl, diffusion_utils.py:10-36; _vb_terms_bpd, :682-713)
//                           : -log P_discretised(x_0 | mean, e^{logvar / 2}) ] / ln 2              (diffusion_utils.py:62-88; tanh CDF approximation :39-44)
//     loss_b = mse_b + vb_b
// In eager PyTorch that is ~100 elementwise / index / reduction launches forward and backward on a (B, 4, 28, 28) tensor -- at the
// reference's one sample per GPU, 6 % of the whole training step.  Here: one launch before the denoiser (x_t) and one after it that
// produces mse, vb, loss per sample AND the gradient of each with respect to the model output (workgroup = sample, fp32
// arithmetic in the reference's order); the backward is the product of that stored gradient with the incoming per-sample scalars.
#include "dm_common.h"

namespace dm {

constexpr int DL_THREADS = 256;

__global__ __launch_bounds__(256) void q_sample_kernel(const dm_training_loss_args p) {
    const int64_t per = (int64_t)p.channels * p.hw;
    const int64_t total = (int64_t)p.batch * per;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / per);
        const int64_t tb = p.t[b];
        // a timestep outside [0, T) (respaced / unspaced mix-up) must not read beyond the tables: the sample is POISONED with NaN so
        // that the loss says so (the generic ATen path trips an index assert there; precondition stated in diffma_hip.h)
        if ((uint64_t)tb >= (uint64_t)p.T) { p.x_t_out[e] = __builtin_nanf(""); continue; }
        const float* tab = p.tables + tb;
        p.x_t_out[e] = tab[(int64_t)p.row_sqrt_ac * p.T] * p.x_start[e] + tab[(int64_t)p.row_sqrt_1mac * p.T] * p.noise[e];
    }
}

// tanh approximation of the standard normal CDF and its derivative (diffusion_utils.py:39-44)
__device__ __forceinline__ void approx_cdf(float x, float& cdf, float& pdf) {
    const float k = 0.7978845608028654f;                      // sqrt(2 / pi)
    const float th = tanhf(k * (x + 0.044715f * x * x * x));
    cdf = 0.5f * (1.0f + th);
    pdf = 0.5f * (1.0f - th * th) * k * (1.0f + 3.0f * 0.044715f * x * x);
}

template <typename TM> __device__ __forceinline__ float rnd(float x) {      // round to the model output's dtype (fp32: identity)
    if constexpr (sizeof(TM) == 4) return x;
    else { TM tmp; io<TM>::st(&tmp, x); return io<TM>::ld(&tmp); }
}

template <typename TM>
__global__ __launch_bounds__(DL_THREADS) void training_loss_kernel(const dm_training_loss_args p) {
    __shared__ float red[2][DL_THREADS / 64];
    const int b = blockIdx.x;
    const int64_t per = (int64_t)p.channels * p.hw;
    const int64_t t = p.t[b];
    if ((uint64_t)t >= (uint64_t)p.T) {                            // out-of-range timestep: NaN loss and gradient for this sample, no table read
        const float qnan = __builtin_nanf("");
        float* Gb = p.grad + (int64_t)b * 2 * per;
        for (int64_t r = threadIdx.x; r < 2 * per; r += DL_THREADS) Gb[r] = qnan;
        if (threadIdx.x == 0) { p.mse[b] = qnan; p.vb[b] = qnan; p.loss[b] = qnan; }
        return;
    }
    const float* tab = p.tables + t;
    const float sr = tab[(int64_t)p.row_sqrt_recip_ac * p.T], srm1 = tab[(int64_t)p.row_sqrt_recipm1_ac * p.T];
    const float minl = tab[(int64_t)p.row_post_logvar * p.T], maxl = tab[(int64_t)p.row_log_betas * p.T];
    const float c1 = tab[(int64_t)p.row_coef1 * p.T], c2 = tab[(int64_t)p.row_coef2 * p.T];
    const float inv_n = 1.0f / (float)per;
    const float inv_ln2 = 1.4426950408889634f;
    const TM* mo = (const TM*)p.model_out + (int64_t)b * 2 * per;
    float* G = p.grad + (int64_t)b * 2 * per;
    float s_mse = 0.0f, s_vb = 0.0f;
    for (int64_t r = threadIdx.x; r < per; r += DL_THREADS) {
        const int64_t e = (int64_t)b * per + r;
        const float eps = io<TM>::ld(mo + r), v = io<TM>::ld(mo + per + r);
        const float x0 = p.x_start[e], xt = p.x_t[e], nz = p.noise[e];
        const float d = nz - eps;
        s_mse += d * d;
        G[r] = -2.0f * d * inv_n;
        // (v + 1) / 2 and 1 - frac are evaluated in the MODEL OUTPUT's dtype by the reference's expression (tensor op scalar keeps the
        // tensor's dtype: under autocast the denoiser returns 16-bit values), everything after meets an fp32 table entry
        const float frac = rnd<TM>(v + 1.0f) * 0.5f;
        const float lv = frac * maxl + rnd<TM>(1.0f - frac) * minl;
        const float px0 = sr * xt - srm1 * eps;
        const float mean = c1 * px0 + c2 * xt, tmean = c1 * x0 + c2 * xt;
        float val, dval;                                         // the bound's term (nats) and its derivative with respect to logvar
        if (t != 0) {
            const float dd = minl - lv, ed = expf(dd), q = (tmean - mean) * (tmean - mean) * expf(-lv);
            val = 0.5f * (-1.0f - dd + ed + q);
            dval = 0.5f * (1.0f - ed - q);
        } else {
            const float inv = expf(-0.5f * lv), cen = x0 - mean;
            const float a = inv * (cen + 1.0f / 255.0f), bb = inv * (cen - 1.0f / 255.0f);
            float ch, ph, cl, pl;
            approx_cdf(a, ch, ph);
            approx_cdf(bb, cl, pl);
            // d a / d logvar = -a / 2,  d b / d logvar = -b / 2;   log(clamp(., 1e-12)): no gradient where the clamp is active
            float arg, darg;
            if (x0 < -0.999f) { arg = ch; darg = ph * (-0.5f * a); }
            else if (x0 > 0.999f) { arg = 1.0f - cl; darg = -pl * (-0.5f * bb); }
            else { arg = ch - cl; darg = ph * (-0.5f * a) - pl * (-0.5f * bb); }
            const bool live = arg >= 1e-12f;
            val = -logf(live ? arg : 1e-12f);
            dval = live ? -darg / arg : 0.0f;
        }
        s_vb += val;
        G[per + r] = dval * 0.5f * (maxl - minl) * inv_n * inv_ln2;       // d logvar / d v = (max_log - min_log) / 2
    }
    // sample sums: wave shuffles, then the workgroup's four waves through LDS
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s_mse += __shfl_down(s_mse, off); s_vb += __shfl_down(s_vb, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s_mse; red[1][wave] = s_vb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f, vb = 0.0f;
#pragma unroll
        for (int w = 0; w < DL_THREADS / 64; ++w) { m += red[0][w]; vb += red[1][w]; }
        m *= inv_n;
        vb *= inv_n * inv_ln2;
        p.mse[b] = m;
        p.vb[b] = vb;
        p.loss[b] = m + vb;
    }
}

// grad_out[b, :C] = g_eps[b] * G[b, :C],  grad_out[b, C:] = g_v[b] * G[b, C:]   (model output dtype)
template <typename TM>
__global__ __launch_bounds__(256) void training_loss_bwd_kernel(const dm_training_loss_args p) {
    const int64_t per = (int64_t)p.channels * p.hw;
    const int64_t total = (int64_t)p.batch * 2 * per;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / (2 * per));
        const bool is_v = (e - (int64_t)b * 2 * per) >= per;
        const float g = is_v ? p.g_v[b] : p.g_eps[b];
        io<TM>::st((TM*)p.grad_out + e, g * p.grad[e]);
    }
}

static int dl_check(const dm_training_loss_args& a, const char* who) {
    if (!a.t || !a.tables) { set_error("%s: null t / tables", who); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.channels <= 0 || a.hw <= 0 || a.T <= 0) { set_error("%s: non-positive size", who); return DM_ERR_ARG; }
    return DM_OK;
}

}  // namespace dm

extern "C" int dm_q_sample(const dm_training_loss_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_q_sample: null args"); return DM_ERR_ARG; }
    const dm_training_loss_args& a = *args;
    if (int rc = dl_check(a, "dm_q_sample")) return rc;
    if (!a.x_start || !a.noise || !a.x_t_out) { set_error("dm_q_sample: null tensor pointer"); return DM_ERR_ARG; }
    const int64_t total = (int64_t)a.batch * a.channels * a.hw;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(q_sample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_q_sample: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

extern "C" int dm_training_loss(const dm_training_loss_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_training_loss: null args"); return DM_ERR_ARG; }
    const dm_training_loss_args& a = *args;
    if (int rc = dl_check(a, "dm_training_loss")) return rc;
    if (!a.model_out || !a.x_start || !a.x_t || !a.noise || !a.mse || !a.vb || !a.loss || !a.grad) { set_error("dm_training_loss: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch > 0x7fffffff / 2) { set_error("dm_training_loss: batch too large"); return DM_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (a.out_dtype) {
        case DM_F32: hipLaunchKernelGGL((training_loss_kernel<float>), dim3(a.batch), dim3(DL_THREADS), 0, st, a); break;
        case DM_BF16: hipLaunchKernelGGL((training_loss_kernel<bf16_t>), dim3(a.batch), dim3(DL_THREADS), 0, st, a); break;
        case DM_F16: hipLaunchKernelGGL((training_loss_kernel<f16_t>), dim3(a.batch), dim3(DL_THREADS), 0, st, a); break;
        default: set_error("dm_training_loss: bad out_dtype %d", a.out_dtype); return DM_ERR_DTYPE;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_training_loss: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}

extern "C" int dm_training_loss_bwd(const dm_training_loss_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_training_loss_bwd: null args"); return DM_ERR_ARG; }
    const dm_training_loss_args& a = *args;
    if (a.batch <= 0 || a.channels <= 0 || a.hw <= 0) { set_error("dm_training_loss_bwd: non-positive size"); return DM_ERR_ARG; }
    if (!a.grad || !a.g_eps || !a.g_v || !a.grad_out) { set_error("dm_training_loss_bwd: null tensor pointer"); return DM_ERR_ARG; }
    const int64_t total = (int64_t)a.batch * 2 * a.channels * a.hw;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    switch (a.out_dtype) {
        case DM_F32: hipLaunchKernelGGL((training_loss_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, a); break;
        case DM_BF16: hipLaunchKernelGGL((training_loss_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, a); break;
        case DM_F16: hipLaunchKernelGGL((training_loss_bwd_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, a); break;
        default: set_error("dm_training_loss_bwd: bad out_dtype %d", a.out_dtype); return DM_ERR_DTYPE;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_training_loss_bwd: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
