"""Timestep respacing (reference diffusion/respace.py:12-129)."""
from __future__ import annotations

import numpy as np
import torch as th

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """Subset of the original steps to keep.  section_counts: list of ints, a comma-separated string, or
    'ddimN' (the fixed integer stride of the DDIM paper that yields exactly N steps)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            steps.append(start + round(pos))
            pos += stride
        start += size
    return set(steps)


class SpacedDiffusion(GaussianDiffusion):
    """A diffusion process that visits only `use_timesteps` of a base process: the betas are re-derived so
    that the kept steps have the same cumulative alphas, and the model is fed the ORIGINAL step numbers."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        self.timestep_map, new_betas, last = [], [], 1.0
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def condition_mean(self, cond_fn, *args, **kwargs):
        return super().condition_mean(self._wrap_model(cond_fn), *args, **kwargs)

    def condition_score(self, cond_fn, *args, **kwargs):
        return super().condition_score(self._wrap_model(cond_fn), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        if not hasattr(self, "_device_maps"):
            self._device_maps = {}                      # shared by every wrapper: one host-to-device copy per (device, dtype), ever
        return _WrappedModel(model, self.timestep_map, self.original_num_steps, self._device_maps)

    def _scale_timesteps(self, t):
        return t


class _WrappedModel:
    """Translates respaced step indices to original ones; the lookup table lives on the device."""

    def __init__(self, model, timestep_map, original_num_steps, device_maps=None):
        self.model, self.timestep_map, self.original_num_steps = model, timestep_map, original_num_steps
        self._maps = device_maps if device_maps is not None else {}

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        m = self._maps.get(key)
        if m is None:
            m = th.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
            self._maps[key] = m
        return self.model(x, m[ts], **kwargs)
