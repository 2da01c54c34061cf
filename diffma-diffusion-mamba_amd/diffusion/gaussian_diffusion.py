"""IDDPM/ADM Gaussian diffusion with the reference's API (diffusion/gaussian_diffusion.py), re-built so
that nothing inside a sampling or training step touches the host:

  * every schedule table is computed once in float64 numpy (public attributes, same names as the
    reference: betas, alphas_cumprod, posterior_variance, ...) and mirrored once per device as ONE fp32
    tensor; `_extract_into_tensor` becomes an on-device row gather (the reference re-uploads a float64
    numpy array on every call, gaussian_diffusion.py:864-875: ~10 H2D copies per sampling step);
  * timestep vectors for the samplers are created on the device (no Python list -> H2D per step).
Numerics are the reference's: tables float64 -> float32 after indexing, all step math in fp32.
"""
from __future__ import annotations

import enum
import os
import math

import numpy as np
import torch as th

from .diffusion_utils import discretized_gaussian_log_likelihood, normal_kl


def mean_flat(tensor):
    """Mean over all non-batch dimensions."""
    return tensor.mean(dim=list(range(1, tensor.dim())))


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()   # model predicts x_{t-1}
    START_X = enum.auto()      # model predicts x_0
    EPSILON = enum.auto()      # model predicts the noise


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()  # interpolate (in log space) between FIXED_SMALL and FIXED_LARGE


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


# ---- beta schedules (reference :58-141) -----------------------------------------------------------
def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    T = num_diffusion_timesteps
    if beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, T, dtype=np.float64)
    elif beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    elif beta_schedule in ("warmup10", "warmup50"):
        frac = 0.1 if beta_schedule == "warmup10" else 0.5
        betas = beta_end * np.ones(T, dtype=np.float64)
        n = int(T * frac)
        betas[:n] = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(T, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (T,)
    return betas


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    T = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta) for i in range(T)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":   # Ho et al., rescaled to any number of steps
        scale = 1000 / num_diffusion_timesteps
        return get_beta_schedule("linear", beta_start=scale * 0.0001, beta_end=scale * 0.02,
                                 num_diffusion_timesteps=num_diffusion_timesteps)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps, lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


_TABLE_NAMES = (
    "betas", "log_betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
    "posterior_mean_coef2", "one_minus_alphas_cumprod", "fixed_large_variance", "fixed_large_log_variance",
)


class GaussianDiffusion:
    """Training and sampling utilities.  `betas` is a 1-D float64 array, one per diffusion step."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type):
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        # posterior q(x_{t-1} | x_t, x_0); log clipped because the variance is 0 at t = 0
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = (
            np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
            if len(self.posterior_variance) > 1 else np.array([]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)
        self._dev_tables = {}

    # ---- device-resident tables ---------------------------------------------------------------------
    def _host_table(self, name):
        if name == "log_betas":
            return np.log(self.betas)
        if name == "one_minus_alphas_cumprod":
            return 1.0 - self.alphas_cumprod
        if name == "fixed_large_variance":
            return np.append(self.posterior_variance[1], self.betas[1:]) if self.num_timesteps > 1 else self.betas
        if name == "fixed_large_log_variance":
            return np.log(self._host_table("fixed_large_variance"))
        arr = getattr(self, name)
        return arr if len(arr) == self.num_timesteps else np.zeros(self.num_timesteps)

    def _tables(self, device):
        key = str(device)
        tab = self._dev_tables.get(key)
        if tab is None:
            host = np.stack([self._host_table(n) for n in _TABLE_NAMES])          # float64 [K, T]
            tab = th.from_numpy(host).to(device=device).float()
            self._dev_tables[key] = tab
        return tab

    def _extract(self, name, t, ndim):
        """Row `name` at timesteps t, shaped (B, 1, ..., 1) for broadcasting against an ndim tensor."""
        row = self._tables(t.device)[_TABLE_NAMES.index(name)]
        return row[t].reshape(t.shape[0], *([1] * (ndim - 1)))

    # ---- forward process ----------------------------------------------------------------------------
    def q_mean_variance(self, x_start, t):
        n = x_start.dim()
        e = lambda name: self._extract(name, t, n).expand(x_start.shape)
        return e("sqrt_alphas_cumprod") * x_start, e("one_minus_alphas_cumprod"), e("log_one_minus_alphas_cumprod")

    def q_sample(self, x_start, t, noise=None):
        """x_t ~ q(x_t | x_0)."""
        if noise is None:
            noise = th.randn_like(x_start)
        assert noise.shape == x_start.shape
        n = x_start.dim()
        return self._extract("sqrt_alphas_cumprod", t, n) * x_start + self._extract("sqrt_one_minus_alphas_cumprod", t, n) * noise

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        n = x_t.dim()
        mean = self._extract("posterior_mean_coef1", t, n) * x_start + self._extract("posterior_mean_coef2", t, n) * x_t
        var = self._extract("posterior_variance", t, n).expand(x_t.shape)
        logvar = self._extract("posterior_log_variance_clipped", t, n).expand(x_t.shape)
        assert mean.shape[0] == var.shape[0] == logvar.shape[0] == x_start.shape[0]
        return mean, var, logvar

    # ---- reverse process ----------------------------------------------------------------------------
    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """Run the model and return {'mean','variance','log_variance','pred_xstart','extra'} of p(x_{t-1}|x_t)."""
        model_kwargs = model_kwargs or {}
        B, C = x.shape[:2]
        assert t.shape == (B,)
        out = model(x, t, **model_kwargs)
        extra = None
        if isinstance(out, tuple):
            out, extra = out
        n = x.dim()
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            assert out.shape == (B, C * 2, *x.shape[2:])
            out, var_values = th.split(out, C, dim=1)
            min_log = self._extract("posterior_log_variance_clipped", t, n)
            max_log = self._extract("log_betas", t, n)
            frac = (var_values + 1) / 2                       # model output in [-1, 1] -> [min_var, max_var]
            log_variance = frac * max_log + (1 - frac) * min_log
            variance = th.exp(log_variance)
        else:
            names = {ModelVarType.FIXED_LARGE: ("fixed_large_variance", "fixed_large_log_variance"),
                     ModelVarType.FIXED_SMALL: ("posterior_variance", "posterior_log_variance_clipped")}[self.model_var_type]
            variance = self._extract(names[0], t, n).expand(x.shape)
            log_variance = self._extract(names[1], t, n).expand(x.shape)

        def finish(x0):
            if denoised_fn is not None:
                x0 = denoised_fn(x0)
            return x0.clamp(-1, 1) if clip_denoised else x0

        if self.model_mean_type == ModelMeanType.START_X:
            pred_xstart = finish(out)
        else:
            pred_xstart = finish(self._predict_xstart_from_eps(x_t=x, t=t, eps=out))
        mean, _, _ = self.q_posterior_mean_variance(x_start=pred_xstart, x_t=x, t=t)
        assert mean.shape == log_variance.shape == pred_xstart.shape == x.shape
        return {"mean": mean, "variance": variance, "log_variance": log_variance, "pred_xstart": pred_xstart, "extra": extra}

    def _predict_xstart_from_eps(self, x_t, t, eps):
        assert x_t.shape == eps.shape
        n = x_t.dim()
        return self._extract("sqrt_recip_alphas_cumprod", t, n) * x_t - self._extract("sqrt_recipm1_alphas_cumprod", t, n) * eps

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        n = x_t.dim()
        return (self._extract("sqrt_recip_alphas_cumprod", t, n) * x_t - pred_xstart) / self._extract("sqrt_recipm1_alphas_cumprod", t, n)

    def condition_mean(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """Sohl-Dickstein et al. (2015) guidance: shift the mean by variance * grad log p(y|x)."""
        grad = cond_fn(x, t, **(model_kwargs or {}))
        return p_mean_var["mean"].float() + p_mean_var["variance"] * grad.float()

    def condition_score(self, cond_fn, p_mean_var, x, t, model_kwargs=None):
        """Song et al. (2020) guidance applied to the score / eps."""
        alpha_bar = self._extract("alphas_cumprod", t, x.dim())
        eps = self._predict_eps_from_xstart(x, t, p_mean_var["pred_xstart"])
        eps = eps - (1 - alpha_bar).sqrt() * cond_fn(x, t, **(model_kwargs or {}))
        out = dict(p_mean_var)
        out["pred_xstart"] = self._predict_xstart_from_eps(x, t, eps)
        out["mean"], _, _ = self.q_posterior_mean_variance(x_start=out["pred_xstart"], x_t=x, t=t)
        return out

    @staticmethod
    def _nonzero_mask(t, ndim):
        return (t != 0).float().view(-1, *([1] * (ndim - 1)))          # no noise at t == 0

    _STEP_ROWS = ("posterior_log_variance_clipped", "log_betas", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_mean_coef1", "posterior_mean_coef2", "alphas_cumprod", "alphas_cumprod_prev")
    # GPU: the step arithmetic after the model call runs as ONE kernel (csrc/diffusion_step.hip); DIFFMA_FUSED_DIFFUSION_STEP=0: ATen
    fused_step = os.environ.get("DIFFMA_FUSED_DIFFUSION_STEP", "1") == "1"

    def _fused_step(self, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, ddim, eta=0.0):
        """The configuration DiffMa samples with (learned-range variance, epsilon prediction, no guidance) on a ROCm device:
        model call + dm_diffusion_step.  Returns None when the generic path has to run."""
        if not (self.fused_step and x.is_cuda and x.dtype == th.float32 and denoised_fn is None and cond_fn is None
                and self.model_var_type == ModelVarType.LEARNED_RANGE and self.model_mean_type == ModelMeanType.EPSILON):
            return None
        from .. import hip_ops
        out = self._wrap_model(model)(x, t, **(model_kwargs or {}))        # SpacedDiffusion maps respaced steps to the original ones
        if isinstance(out, tuple) or out.shape != (x.shape[0], 2 * x.shape[1], *x.shape[2:]) or out.dtype not in (th.float32, th.bfloat16, th.float16):
            raise ValueError("the denoiser must return one (B, 2C, ...) tensor with learn_sigma (reference gaussian_diffusion.py:286)")
        noise = th.randn_like(x)
        rows = [_TABLE_NAMES.index(n) for n in self._STEP_ROWS]
        sample, x0 = hip_ops.diffusion_step(out, x, t, noise, self._tables(x.device), rows, ddim=ddim, eta=eta, clip=clip_denoised)
        return {"sample": sample, "pred_xstart": x0}

    def _wrap_model(self, model):
        return model               # overridden by SpacedDiffusion (timestep translation)

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """One ancestral step x_t -> x_{t-1}: {'sample', 'pred_xstart'}."""
        fused = self._fused_step(model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, ddim=False)
        if fused is not None:
            return fused
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        noise = th.randn_like(x)
        if cond_fn is not None:
            out["mean"] = self.condition_mean(cond_fn, out, x, t, model_kwargs=model_kwargs)
        sample = out["mean"] + self._nonzero_mask(t, x.dim()) * th.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def _loop(self, step_fn, model, shape, noise, device, progress, **kw):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else th.randn(*shape, device=device)
        steps = range(self.num_timesteps - 1, -1, -1)
        if progress:
            from tqdm.auto import tqdm   # lazy: tqdm is optional
            steps = tqdm(steps)
        for i in steps:
            t = th.full((shape[0],), i, device=device, dtype=th.long)
            with th.no_grad():
                out = step_fn(model, img, t, **kw)
                yield out
                img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        yield from self._loop(self.p_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised,
                              denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False):
        """Full ancestral sampling; returns the final batch of samples."""
        final = None
        for final in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                    denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                    device=device, progress=progress):
            pass
        return final["sample"]

    # ---- DDIM ------------------------------------------------------------------------------------------
    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        fused = self._fused_step(model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, ddim=True, eta=eta)
        if fused is not None:
            return fused
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        if cond_fn is not None:
            out = self.condition_score(cond_fn, out, x, t, model_kwargs=model_kwargs)
        eps = self._predict_eps_from_xstart(x, t, out["pred_xstart"])      # re-derived, valid for any mean type
        n = x.dim()
        ab, ab_prev = self._extract("alphas_cumprod", t, n), self._extract("alphas_cumprod_prev", t, n)
        sigma = eta * th.sqrt((1 - ab_prev) / (1 - ab)) * th.sqrt(1 - ab / ab_prev)
        noise = th.randn_like(x)
        mean_pred = out["pred_xstart"] * th.sqrt(ab_prev) + th.sqrt(1 - ab_prev - sigma ** 2) * eps   # DDIM eq. 12
        sample = mean_pred + self._nonzero_mask(t, n) * sigma * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0):
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        if cond_fn is not None:
            out = self.condition_score(cond_fn, out, x, t, model_kwargs=model_kwargs)
        n = x.dim()
        eps = (self._extract("sqrt_recip_alphas_cumprod", t, n) * x - out["pred_xstart"]) / self._extract("sqrt_recipm1_alphas_cumprod", t, n)
        ab_next = self._extract("alphas_cumprod_next", t, n)
        return {"sample": out["pred_xstart"] * th.sqrt(ab_next) + th.sqrt(1 - ab_next) * eps, "pred_xstart": out["pred_xstart"]}

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0):
        yield from self._loop(self.ddim_sample, model, shape, noise, device, progress, clip_denoised=clip_denoised,
                              denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0):
        final = None
        for final in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                       denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                       device=device, progress=progress, eta=eta):
            pass
        return final["sample"]

    # ---- losses ----------------------------------------------------------------------------------------
    def _vb_terms_bpd(self, model, x_start, x_t, t, clip_denoised=True, model_kwargs=None):
        """One term of the variational bound in bits/dim: decoder NLL at t == 0, KL otherwise."""
        true_mean, _, true_logvar = self.q_posterior_mean_variance(x_start=x_start, x_t=x_t, t=t)
        out = self.p_mean_variance(model, x_t, t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
        kl = mean_flat(normal_kl(true_mean, true_logvar, out["mean"], out["log_variance"])) / math.log(2.0)
        nll = -discretized_gaussian_log_likelihood(x_start, means=out["mean"], log_scales=0.5 * out["log_variance"])
        assert nll.shape == x_start.shape
        nll = mean_flat(nll) / math.log(2.0)
        return {"output": th.where(t == 0, nll, kl), "pred_xstart": out["pred_xstart"]}

    _LOSS_ROWS = ("sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_log_variance_clipped", "log_betas",
                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_mean_coef1", "posterior_mean_coef2")
    # GPU: q_sample and everything after the denoiser call (mse, the bound's term, their gradients) as two kernels
    # (csrc/diffusion_loss.hip); DIFFMA_FUSED_LOSS=0: the ATen chain below
    fused_loss = os.environ.get("DIFFMA_FUSED_LOSS", "1") == "1"

    def _fused_training_losses(self, model, x_start, t, model_kwargs, noise):
        """The configuration DiffMa trains with (epsilon prediction, learned-range variance, MSE loss) on a ROCm device; None when the
        generic path has to run."""
        if not (self.fused_loss and x_start.is_cuda and x_start.dtype == th.float32 and noise.dtype == th.float32 and t.dtype == th.int64
                and self.loss_type == LossType.MSE and self.model_var_type == ModelVarType.LEARNED_RANGE
                and self.model_mean_type == ModelMeanType.EPSILON and x_start.dim() >= 3):
            return None
        if th.is_grad_enabled() and (x_start.requires_grad or noise.requires_grad):
            return None                         # the kernels give the gradient with respect to the MODEL OUTPUT only (what training needs)
        from .. import hip_ops
        tables = self._tables(x_start.device)
        rows = [_TABLE_NAMES.index(n) for n in self._LOSS_ROWS]
        with th.no_grad():
            x_t = hip_ops.q_sample(x_start, noise, t, tables, rows)
        out = model(x_t, t, **model_kwargs)
        if isinstance(out, tuple) or out.shape != (x_start.shape[0], 2 * x_start.shape[1], *x_start.shape[2:]) \
                or out.dtype not in (th.float32, th.bfloat16, th.float16):
            raise ValueError("the denoiser must return one (B, 2C, ...) tensor with learn_sigma (reference gaussian_diffusion.py:286)")
        mse, vb, loss = _FusedLossFn.apply(out, x_start.contiguous(), x_t, noise.contiguous(), t, tables, rows)
        return {"mse": mse, "vb": vb, "loss": loss}

    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """{'loss' [N], ...}: MSE on the mean prediction (+ 'vb' when the variance is learned)."""
        model_kwargs = model_kwargs or {}
        if noise is None:
            noise = th.randn_like(x_start)
        fused = self._fused_training_losses(model, x_start, t, model_kwargs, noise)
        if fused is not None:
            return fused
        x_t = self.q_sample(x_start, t, noise=noise)
        terms = {}
        if self.loss_type.is_vb():
            terms["loss"] = self._vb_terms_bpd(model=model, x_start=x_start, x_t=x_t, t=t, clip_denoised=False,
                                               model_kwargs=model_kwargs)["output"]
            if self.loss_type == LossType.RESCALED_KL:
                terms["loss"] = terms["loss"] * self.num_timesteps
            return terms
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError(self.loss_type)
        out = model(x_t, t, **model_kwargs)
        if self.model_var_type in (ModelVarType.LEARNED, ModelVarType.LEARNED_RANGE):
            B, C = x_t.shape[:2]
            assert out.shape == (B, C * 2, *x_t.shape[2:])
            out, var_values = th.split(out, C, dim=1)
            # the bound trains the variance only: the mean prediction enters it detached
            frozen = th.cat([out.detach(), var_values], dim=1)
            terms["vb"] = self._vb_terms_bpd(model=lambda *a, r=frozen: r, x_start=x_start, x_t=x_t, t=t,
                                             clip_denoised=False)["output"]
            if self.loss_type == LossType.RESCALED_MSE:
                terms["vb"] = terms["vb"] * (self.num_timesteps / 1000.0)
        if self.model_mean_type == ModelMeanType.PREVIOUS_X:
            target = self.q_posterior_mean_variance(x_start=x_start, x_t=x_t, t=t)[0]
        elif self.model_mean_type == ModelMeanType.START_X:
            target = x_start
        else:
            target = noise
        assert out.shape == target.shape == x_start.shape
        terms["mse"] = mean_flat((target - out) ** 2)
        terms["loss"] = terms["mse"] + terms["vb"] if "vb" in terms else terms["mse"]
        return terms

    def _prior_bpd(self, x_start):
        t = th.full((x_start.shape[0],), self.num_timesteps - 1, device=x_start.device, dtype=th.long)
        mean, _, logvar = self.q_mean_variance(x_start, t)
        return mean_flat(normal_kl(mean1=mean, logvar1=logvar, mean2=0.0, logvar2=0.0)) / math.log(2.0)

    def calc_bpd_loop(self, model, x_start, clip_denoised=True, model_kwargs=None):
        """Whole variational bound, term by term (bits/dim)."""
        vb, xstart_mse, mse = [], [], []
        for step in range(self.num_timesteps - 1, -1, -1):
            t = th.full((x_start.shape[0],), step, device=x_start.device, dtype=th.long)
            noise = th.randn_like(x_start)
            x_t = self.q_sample(x_start=x_start, t=t, noise=noise)
            with th.no_grad():
                out = self._vb_terms_bpd(model, x_start=x_start, x_t=x_t, t=t, clip_denoised=clip_denoised, model_kwargs=model_kwargs)
            vb.append(out["output"])
            xstart_mse.append(mean_flat((out["pred_xstart"] - x_start) ** 2))
            mse.append(mean_flat((self._predict_eps_from_xstart(x_t, t, out["pred_xstart"]) - noise) ** 2))
        vb, xstart_mse, mse = th.stack(vb, dim=1), th.stack(xstart_mse, dim=1), th.stack(mse, dim=1)
        prior = self._prior_bpd(x_start)
        return {"total_bpd": vb.sum(dim=1) + prior, "prior_bpd": prior, "vb": vb, "xstart_mse": xstart_mse, "mse": mse}


class _FusedLossFn(th.autograd.Function):
    """(mse, vb, loss) per sample from the model output; the forward launch also leaves d(term)/d(model output), the backward scales it
    by the incoming per-sample gradients (loss = mse + vb: its gradient goes to both halves)."""

    @staticmethod
    def forward(ctx, out, x_start, x_t, noise, t, tables, rows):
        from .. import hip_ops
        mse, vb, loss, grad = hip_ops.training_loss_fwd(out.contiguous(), x_start, x_t, noise, t, tables, rows)
        ctx.save_for_backward(grad)
        ctx.out_dtype = out.dtype
        return mse, vb, loss

    @staticmethod
    def backward(ctx, g_mse, g_vb, g_loss):
        from .. import hip_ops
        (grad,) = ctx.saved_tensors
        zero = None
        def total(a, b):
            nonlocal zero
            if a is None and b is None:
                zero = grad.new_zeros(grad.shape[0]) if zero is None else zero
                return zero
            return a if b is None else b if a is None else a + b
        return hip_ops.training_loss_bwd(grad, total(g_mse, g_loss), total(g_vb, g_loss), ctx.out_dtype), None, None, None, None, None, None


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """Reference-compatible helper (gaussian_diffusion.py:864-875) for external callers; the class itself
    uses the device-resident tables instead."""
    res = th.from_numpy(np.asarray(arr)).to(device=timesteps.device)[timesteps].float()
    while res.dim() < len(broadcast_shape):
        res = res[..., None]
    return res + th.zeros(broadcast_shape, device=timesteps.device)
