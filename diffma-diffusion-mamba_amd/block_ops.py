"""Autograd wrappers of the fused block elementwise kernels (csrc/block_ops.hip).

`ln_modulate_mask` :  LN(x) -> modulate(shift, scale) -> (x_ssm, x_ssm * w)      (block/mamba_block.py:103-105)
`ln_cat`           :  LN(cat[xs, ws])                                             (attention_network[0], :90,111)
`blend_residual`   :  x + gate * (a*xs + (1-a)*ws)                                (:113-114)
Each is one HBM pass; outputs that feed a GEMM are produced directly in the autocast dtype, so the chain of
ATen kernels (layer_norm, mul, add, cat, copy/cast) of the eager formulation disappears.
"""
from __future__ import annotations

import torch

from . import hip_ops


def _autocast_dtype(x):
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype


class _LnModFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, gamma, beta, shift, scale, mask, eps, y_dtype):
        x = x.contiguous()
        x2c = x2.contiguous() if x2 is not None else None
        y1, y2, stats = hip_ops.ln_mod_fwd(x, x2c, gamma, beta, shift, scale, mask, eps, y_dtype)
        ctx.eps = eps
        ctx.has = (x2 is not None, gamma is not None, beta is not None, scale is not None, mask is not None)
        ctx.save_for_backward(x, x2c, gamma, beta, shift, scale, mask, stats)
        if mask is not None:
            return y1, y2
        return y1

    @staticmethod
    def backward(ctx, dy1, dy2=None):
        x, x2, gamma, beta, shift, scale, mask, stats = ctx.saved_tensors
        has_x2, has_g, has_b, has_mod, has_mask = ctx.has
        if dy1 is None:
            dy1 = torch.zeros((x.shape[0], x.shape[1], x.shape[2] + (x2.shape[2] if has_x2 else 0)), dtype=dy2.dtype, device=x.device)
        dy1 = dy1.contiguous()
        if dy2 is not None:
            dy2 = dy2.contiguous().to(dy1.dtype)
        dx, dx2, dshift, dscale, dgamma, dbeta = hip_ops.ln_mod_bwd(x, x2, gamma, beta, shift, scale, mask, ctx.eps, stats, dy1,
                                                                    dy2 if has_mask else None)
        return (dx, dx2 if has_x2 else None, dgamma.to(gamma.dtype) if has_g else None, dbeta.to(beta.dtype) if has_b else None,
                dshift.to(shift.dtype) if has_mod else None, dscale.to(scale.dtype) if has_mod else None, None, None, None)


def ln_modulate_mask(x, norm: torch.nn.LayerNorm, shift, scale, w):
    """(modulate(LN(x), shift, scale), same * w) with outputs in the autocast dtype.  shift/scale: [B, C] views."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _LnModFn.apply(x, None, norm.weight, norm.bias, shift, scale, w.reshape(x.shape[0], x.shape[1]).contiguous().to(shift.dtype),
                              norm.eps, _autocast_dtype_cached[0])


def ln_cat(xs, ws, norm: torch.nn.LayerNorm):
    """LN(cat[xs, ws], dim=-1) without materialising the cat; output in the dtype of xs."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _LnModFn.apply(xs, ws, norm.weight, norm.bias, None, None, None, norm.eps, xs.dtype)


class _BlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xs, ws, a_row, gate):
        x, xs, ws, a_row = x.contiguous(), xs.contiguous(), ws.contiguous(), a_row.contiguous()
        out = hip_ops.blend_fwd(x, xs, ws, a_row, gate)
        ctx.save_for_backward(xs, ws, a_row, gate)
        return out

    @staticmethod
    def backward(ctx, g):
        xs, ws, a_row, gate = ctx.saved_tensors
        g = g.contiguous()
        dxs, dws, da, dgate = hip_ops.blend_bwd(g, xs, ws, a_row, gate)
        return g, dxs, dws, da, dgate.to(gate.dtype)


def blend_residual(x, xs, ws, a_row, gate):
    """x + gate[:, None] * (a*xs + (1-a)*ws);  a_row [B, L, 1] in the dtype of xs, gate [B, C] view."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _BlendFn.apply(x, xs, ws, a_row.to(xs.dtype), gate)


_autocast_dtype_cached = [torch.float32]


def set_output_dtype(dtype):
    """Called by the block right before ln_modulate_mask (autocast state must be read outside the disabled region)."""
    _autocast_dtype_cached[0] = dtype
