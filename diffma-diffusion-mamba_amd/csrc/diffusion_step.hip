// K9  dm_diffusion_step -- one reverse-diffusion step of the sampler after the denoiser call, in ONE elementwise pass.
//
// Replaces the chain of ~25 tiny ATen kernels behind GaussianDiffusion.p_sample / ddim_sample for the configuration DiffMa uses
// (learned-range variance, epsilon prediction; reference diffusion/gaussian_diffusion.py:285-323 p_mean_variance, :232-252
// q_posterior_mean_variance, :410-416 p_sample, :548-598 ddim_sample):
//     eps, v   = model_out[:, :C], model_out[:, C:]
//     logvar   = frac * log(beta_t) + (1 - frac) * log(posterior_var_t),   frac = (v + 1) / 2
//     x0       = sqrt(1/abar_t) * x - sqrt(1/abar_t - 1) * eps            (clamped to [-1, 1] if clip)
//     DDPM:  sample = coef1_t * x0 + coef2_t * x + [t != 0] * exp(logvar / 2) * noise
//     DDIM:  eps'   = (sqrt(1/abar_t) * x - x0) / sqrt(1/abar_t - 1)
//            sigma  = eta * sqrt((1 - abar_prev) / (1 - abar_t)) * sqrt(1 - abar_t / abar_prev)
//            sample = x0 * sqrt(abar_prev) + sqrt(1 - abar_prev - sigma^2) * eps' + [t != 0] * sigma * noise
// The per-timestep coefficients come from the caller's device-resident fp32 table [nrows][T] (row numbers in the args); t is
// read per batch element, so one launch serves a batch of mixed timesteps.  Pure HBM work: 5 reads + 2 writes per element.
#include "dm_common.h"

namespace dm {

template <typename TM>
__global__ __launch_bounds__(256) void diffusion_step_kernel(const dm_diffusion_step_args p) {
    const int64_t per = (int64_t)p.channels * p.hw;                   // elements per sample (of x)
    const int64_t total = (int64_t)p.batch * per;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / per);
        const int64_t r = e - (int64_t)b * per;
        const int64_t t = p.t[b];
        const float* tab = p.tables + t;
        const float c_recip = tab[(int64_t)p.row_sqrt_recip_ac * p.T], c_recipm1 = tab[(int64_t)p.row_sqrt_recipm1_ac * p.T];
        const float min_log = tab[(int64_t)p.row_post_logvar * p.T], max_log = tab[(int64_t)p.row_log_betas * p.T];
        const float eps = io<TM>::ld((const TM*)p.model_out + (int64_t)b * 2 * per + r);
        const float v = io<TM>::ld((const TM*)p.model_out + (int64_t)b * 2 * per + per + r);
        const float x = p.x[e], nz = p.noise ? p.noise[e] : 0.0f;
        const float frac = (v + 1.0f) * 0.5f;
        const float logvar = frac * max_log + (1.0f - frac) * min_log;
        float x0 = c_recip * x - c_recipm1 * eps;
        if (p.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        const float nzmask = (t != 0) ? 1.0f : 0.0f;
        float sample;
        if (p.mode == 0) {
            const float mean = tab[(int64_t)p.row_coef1 * p.T] * x0 + tab[(int64_t)p.row_coef2 * p.T] * x;
            sample = mean + nzmask * expf(0.5f * logvar) * nz;
        } else {
            const float ab = tab[(int64_t)p.row_ac * p.T], abp = tab[(int64_t)p.row_ac_prev * p.T];
            const float eps2 = (c_recip * x - x0) / c_recipm1;
            const float sigma = p.eta * sqrtf((1.0f - abp) / (1.0f - ab)) * sqrtf(1.0f - ab / abp);
            sample = x0 * sqrtf(abp) + sqrtf(1.0f - abp - sigma * sigma) * eps2 + nzmask * sigma * nz;
        }
        p.sample[e] = sample;
        if (p.pred_xstart) p.pred_xstart[e] = x0;
    }
}

}  // namespace dm

extern "C" int dm_diffusion_step(const dm_diffusion_step_args* args, void* stream) {
    using namespace dm;
    if (!args) { set_error("dm_diffusion_step: null args"); return DM_ERR_ARG; }
    const dm_diffusion_step_args& a = *args;
    if (!a.model_out || !a.x || !a.t || !a.tables || !a.sample) { set_error("dm_diffusion_step: null tensor pointer"); return DM_ERR_ARG; }
    if (a.batch <= 0 || a.channels <= 0 || a.hw <= 0 || a.T <= 0) { set_error("dm_diffusion_step: non-positive size"); return DM_ERR_ARG; }
    if (a.mode != 0 && a.mode != 1) { set_error("dm_diffusion_step: mode must be 0 (DDPM) or 1 (DDIM)"); return DM_ERR_ARG; }
    const int64_t total = (int64_t)a.batch * a.channels * a.hw;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipStream_t st = (hipStream_t)stream;
    switch (a.out_dtype) {
        case DM_F32: hipLaunchKernelGGL((diffusion_step_kernel<float>), dim3(blocks), dim3(256), 0, st, a); break;
        case DM_BF16: hipLaunchKernelGGL((diffusion_step_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, a); break;
        case DM_F16: hipLaunchKernelGGL((diffusion_step_kernel<f16_t>), dim3(blocks), dim3(256), 0, st, a); break;
        default: set_error("dm_diffusion_step: bad out_dtype %d", a.out_dtype); return DM_ERR_DTYPE;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dm_diffusion_step: launch failed: %s", hipGetErrorString(e)); return DM_ERR_LAUNCH; }
    return DM_OK;
}
