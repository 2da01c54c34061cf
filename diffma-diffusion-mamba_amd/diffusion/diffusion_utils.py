"""Gaussian helpers for the variational-bound term (reference diffusion/diffusion_utils.py:10-88)."""
from __future__ import annotations

import math

import torch as th


def _as_tensor(v, like):
    return v if isinstance(v, th.Tensor) else th.tensor(v).to(like)


def normal_kl(mean1, logvar1, mean2, logvar2):
    """KL( N(mean1, e^logvar1) || N(mean2, e^logvar2) ), broadcasting; at least one argument is a Tensor."""
    like = next((o for o in (mean1, logvar1, mean2, logvar2) if isinstance(o, th.Tensor)), None)
    assert like is not None, "at least one argument must be a Tensor"
    logvar1, logvar2 = _as_tensor(logvar1, like), _as_tensor(logvar2, like)
    d = logvar1 - logvar2
    return 0.5 * (-1.0 - d + th.exp(d) + (mean1 - mean2) ** 2 * th.exp(-logvar2))


def approx_standard_normal_cdf(x):
    """tanh approximation of the standard normal CDF."""
    return 0.5 * (1.0 + th.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * th.pow(x, 3))))


def continuous_gaussian_log_likelihood(x, *, means, log_scales):
    z = (x - means) * th.exp(-log_scales)
    return -0.5 * z * z - 0.5 * math.log(2.0 * math.pi)


def discretized_gaussian_log_likelihood(x, *, means, log_scales):
    """log-likelihood of x (uint8 images rescaled to [-1, 1]) under a Gaussian discretised to 1/255 bins."""
    assert x.shape == means.shape == log_scales.shape
    centered = x - means
    inv_std = th.exp(-log_scales)
    cdf_hi = approx_standard_normal_cdf(inv_std * (centered + 1.0 / 255.0))
    cdf_lo = approx_standard_normal_cdf(inv_std * (centered - 1.0 / 255.0))
    log_hi = th.log(cdf_hi.clamp(min=1e-12))
    log_one_minus_lo = th.log((1.0 - cdf_lo).clamp(min=1e-12))
    log_mid = th.log((cdf_hi - cdf_lo).clamp(min=1e-12))
    out = th.where(x < -0.999, log_hi, th.where(x > 0.999, log_one_minus_lo, log_mid))
    assert out.shape == x.shape
    return out
