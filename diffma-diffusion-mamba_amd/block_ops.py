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


# PASS-THROUGH outputs.  x (and x2) of the two LayerNorm nodes of a block have a SECOND consumer -- the residual / the blend -- so
# autograd would add the two gradient contributions with one elementwise pass per tensor (3 per block: 48 launches, 3.1 ms per
# DiffMa-L/2 step at batch 512).  With passthrough=True the node also returns x (x2) itself; the caller hands THAT alias to the
# other consumer, the input then has one consumer only, and this node's backward receives the other consumer's gradient and lets
# dm_ln_mod_bwd add it (one extra read inside a pass that runs anyway).  Two forms: the blend's gradients of x_ssm / w_ssm are
# fresh tensors of its own backward and are accumulated IN PLACE; the residual's gradient is the block output's gradient itself
# -- autograd may share that tensor with other nodes (an AddBackward of a long skip hands ONE tensor to both operands) -- so it
# is only READ (dx_add) and the sum goes to a new tensor.
PASSTHROUGH = __import__("os").environ.get("DIFFMA_LN_PASSTHROUGH", "1") == "1"


class _LnModFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x2, gamma, beta, shift, scale, mask, eps, y_dtype, passthrough=False):
        xc = x.contiguous()
        x2c = x2.contiguous() if x2 is not None else None
        y1, y2, stats = hip_ops.ln_mod_fwd(xc, x2c, gamma, beta, shift, scale, mask, eps, y_dtype)
        ctx.eps = eps
        ctx.has = (x2 is not None, gamma is not None, beta is not None, scale is not None, mask is not None)
        ctx.passthrough = passthrough
        ctx.save_for_backward(xc, x2c, gamma, beta, shift, scale, mask, stats)
        outs = (y1, y2) if mask is not None else (y1,)
        if passthrough:
            outs = outs + ((x, x2) if x2 is not None else (x,))
        return outs if len(outs) > 1 else outs[0]

    @staticmethod
    def backward(ctx, *grads):
        x, x2, gamma, beta, shift, scale, mask, stats = ctx.saved_tensors
        has_x2, has_g, has_b, has_mod, has_mask = ctx.has
        ny = 2 if has_mask else 1
        dy1, dy2 = grads[0], (grads[1] if has_mask else None)
        res = grads[ny:] if ctx.passthrough else ()
        if dy1 is None:
            dy1 = torch.zeros((x.shape[0], x.shape[1], x.shape[2] + (x2.shape[2] if has_x2 else 0)), dtype=dy2.dtype, device=x.device)
        dy1 = dy1.contiguous()
        if dy2 is not None:
            dy2 = dy2.contiguous().to(dy1.dtype)
        # gradients that arrived through the pass-through aliases
        rx = res[0] if len(res) > 0 else None
        rx2 = res[1] if len(res) > 1 else None
        kw, plain = {}, True
        if has_x2 and rx is not None and rx2 is not None:               # LN(cat): the blend's own fresh dxs / dws -> in place
            rx = rx if (rx.is_contiguous() and rx.dtype == x.dtype) else rx.contiguous().to(x.dtype)
            rx2 = rx2 if (rx2.is_contiguous() and rx2.dtype == x2.dtype) else rx2.contiguous().to(x2.dtype)
            kw, plain = dict(dx=rx, dx2=rx2, accumulate=True), False
        elif not has_x2 and rx is not None and rx.dtype == x.dtype and rx.is_contiguous():
            kw, plain = dict(dx_add=rx), False                          # residual: possibly shared -> read only
        if has_mod:
            kw["mod_dtype"] = shift.dtype
        dx, dx2, dshift, dscale, dgamma, dbeta = hip_ops.ln_mod_bwd(x, x2, gamma, beta, shift, scale, mask, ctx.eps, stats, dy1,
                                                                    dy2 if has_mask else None, **kw)
        if plain:                                     # odd cases (one alias unused, dtype mismatch): plain sums
            if rx is not None:
                dx = dx + rx.to(dx.dtype)
            if rx2 is not None:
                dx2 = dx2 + rx2.to(dx2.dtype)
        return (dx, dx2 if has_x2 else None, dgamma.to(gamma.dtype) if has_g else None, dbeta.to(beta.dtype) if has_b else None,
                dshift.to(shift.dtype) if has_mod else None, dscale.to(scale.dtype) if has_mod else None, None, None, None, None)


import threading

_MASK_TLS = threading.local()          # per host thread: two threads driving forwards on two streams never share a copy


def drop_mask_cache():
    """Forget the cached soft-mask copy of this thread.  Called by DiffMa.forward at the end of a forward that ran inside a stream
    capture (the copy lives in THAT graph's private pool: a later capture with the same mask object must make its own) and by
    step_prep.invalidate() (writers that do not bump `_version`: graph replays, `.data` assignments)."""
    _MASK_TLS.ent = None


def _mask_once(w, B, L, dtype):
    """The soft mask as a contiguous [B, L] tensor in the modulation dtype: every block of the denoiser applies the SAME mask
    (reference block/mamba_block.py:103), so it is reshaped and cast once per mask tensor instead of once per block (a cast launch
    per block and step at the reference's batch).  Keyed on the tensor object, its version, step_prep's generation counter and the
    capture state; held weakly; thread-local."""
    import weakref

    from . import step_prep

    ent = getattr(_MASK_TLS, "ent", None)
    grad = w.requires_grad and torch.is_grad_enabled()
    # a copy made outside a stream capture must not be baked into a captured graph (it is freed when the next mask replaces it) and a
    # copy made inside one lives in that graph's pool (and is dropped when that forward ends): the capture state is part of the key
    cap = w.is_cuda and torch.cuda.is_current_stream_capturing()
    gen = step_prep._GEN[0]
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == dtype and ent[3].shape == (B, L) and ent[4] == cap \
            and ent[5] == gen and not grad:
        return ent[3]
    m = w.reshape(B, L).contiguous().to(dtype)
    if not grad:
        _MASK_TLS.ent = (weakref.ref(w), w._version, dtype, m, cap, gen)
    return m


def ln_modulate_mask(x, norm: torch.nn.LayerNorm, shift, scale, w, passthrough=False):
    """(modulate(LN(x), shift, scale), same * w) with outputs in the autocast dtype.  shift/scale: [B, C] views.
    passthrough: also returns x itself for the residual branch (see PASS-THROUGH above): (x_ssm, w_ssm, x_res)."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _LnModFn.apply(x, None, norm.weight, norm.bias, shift, scale, _mask_once(w, x.shape[0], x.shape[1], shift.dtype),
                              norm.eps, _autocast_dtype_cached[0], passthrough)


def ln_cat(xs, ws, norm: torch.nn.LayerNorm, passthrough=False):
    """LN(cat[xs, ws], dim=-1) without materialising the cat; output in the dtype of xs.
    passthrough: (hcat, xs, ws) with the two aliases to be used by the blend."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _LnModFn.apply(xs, ws, norm.weight, norm.bias, None, None, None, norm.eps, xs.dtype, passthrough)


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
        return g, dxs, dws, da, dgate


def blend_residual(x, xs, ws, a_row, gate):
    """x + gate[:, None] * (a*xs + (1-a)*ws);  a_row [B, L, 1] in the dtype of xs, gate [B, C] view."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _BlendFn.apply(x, xs, ws, a_row.to(xs.dtype), gate)


class _GateHeadFn(torch.autograd.Function):
    """a = sigmoid(silu(h + b1) @ w2^T + b2) with h the first Linear's output without its bias (csrc/gate_head.hip)."""

    @staticmethod
    def forward(ctx, h, b1, w2, b2):
        h = h if h.stride(-1) == 1 else h.contiguous()
        b1f = None if b1 is None else b1.detach().float().contiguous()
        w2f = w2.detach().float().reshape(-1).contiguous()
        b2f = None if b2 is None else b2.detach().float().reshape(-1).contiguous()
        a = hip_ops.gate_head_fwd(h, b1f, w2f, b2f)
        ctx.save_for_backward(h, b1f, w2f, a)
        ctx.meta = (None if b1 is None else b1.dtype, w2.dtype, w2.shape, None if b2 is None else (b2.dtype, b2.shape))
        return a

    @staticmethod
    def backward(ctx, da):
        h, b1f, w2f, a = ctx.saved_tensors
        b1_dt, w2_dt, w2_shape, b2_meta = ctx.meta
        dh, db1, dw2, db2 = hip_ops.gate_head_bwd(da, a, h, b1f, w2f)
        return (dh, None if b1_dt is None else db1.to(b1_dt), dw2.to(w2_dt).view(w2_shape),
                None if b2_meta is None else db2.to(b2_meta[0]).view(b2_meta[1]))


# DIFFMA_GATE_HEAD=0: the ATen chain (bias epilogue, SiLU, Linear(C, 1), Sigmoid) instead of the fused tail (A/B runs, tests)
GATE_HEAD = __import__("os").environ.get("DIFFMA_GATE_HEAD", "1") == "1"


def gate_head(h, b1, w2, b2):
    """sigmoid(Linear(C, 1)(silu(h + b1))): the tail of the block's fusion MLP; h [B, L, C] -> [B, L, 1]."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _GateHeadFn.apply(h, b1, w2, b2)


_autocast_dtype_cached = [torch.float32]


def set_output_dtype(dtype):
    """Called by the block right before ln_modulate_mask (autocast state must be read outside the disabled region)."""
    _autocast_dtype_cached[0] = dtype
