"""fp64 CPU restatement of the Mamba-2 operator used by `--use-mamba2`.  TEST INFRASTRUCTURE ONLY.

The reference calls `mamba_split_conv1d_scan_combined` from the absent wheel mamba-ssm==2.0.4 (block/mamba2.py:17-21; call
sites :392-450).  This restates its published semantics sequentially: split [z | xBC | dt], causal conv + SiLU on
xBC, per-head scalar-decay state-space recurrence (one chunk, since chunk_size 256 >= L), gated RMSNorm, out_proj.

PARITY PINNED (round 2) by the reference's own pure-PyTorch `Mamba2.step()` (block/mamba2.py:715-775) run token by token in
fp64 (G10): conv, recurrence, D skip and the SiLU(z) gate are reference-held end to end (cases m2.a / m2.b, rmsnorm=False);
with rmsnorm=True (m2.c / m2.d) the reference calls the absent wheel's RMSNormGated, for which the generator supplies the
documented forward (norm_before_gate=False: rmsnorm(y * silu(z)) * weight).  tests/test_golden_cpu.py holds
`mamba_split_conv1d_scan_combined_ref` to G10 at <= 1e-9.
"""
from __future__ import annotations

import torch

from .mamba_ref import causal_conv1d_ref, softplus_ref


def rmsnorm_gated_ref(y, z, weight, eps, norm_before_gate=False, dtype=torch.float64):
    """norm_before_gate=False (DiffMa, block/mamba2.py:248,349): RMSNorm(y * silu(z)) * w over the last dim."""
    y, z, w = y.to(dtype), z.to(dtype), weight.to(dtype)
    if norm_before_gate:
        yn = y * torch.rsqrt(y.pow(2).mean(-1, keepdim=True) + eps) * w
        return yn * (z * torch.sigmoid(z))
    g = y * (z * torch.sigmoid(z))
    return g * torch.rsqrt(g.pow(2).mean(-1, keepdim=True) + eps) * w


def ssd_scan_ref(x, dt, A, Bm, Cm, D, headdim, dtype=torch.float64):
    """x: (B, L, H*P); dt: (B, L, H) (already softplus'd); A: (H,); Bm, Cm: (B, L, N); D: (H,).
    S_l = exp(dt_l A_h) S_{l-1} + dt_l x_l (outer) B_l ;  y_l = S_l C_l + D_h x_l."""
    x, dt, A, Bm, Cm, D = (t.to(dtype) for t in (x, dt, A, Bm, Cm, D))
    Bsz, L, dim = x.shape
    H = dim // headdim
    xh = x.view(Bsz, L, H, headdim)
    S = torch.zeros(Bsz, H, headdim, Bm.shape[-1], dtype=dtype, device=x.device)
    ys = []
    for l in range(L):
        a = torch.exp(dt[:, l] * A[None])                                        # (B, H)
        S = a[:, :, None, None] * S + (dt[:, l, :, None] * xh[:, l])[..., None] * Bm[:, l, None, None, :]
        ys.append((S * Cm[:, l, None, None, :]).sum(-1) + D[None, :, None] * xh[:, l])
    return torch.stack(ys, dim=1).reshape(Bsz, L, dim)


def mamba_split_conv1d_scan_combined_ref(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size=256, initial_states=None,
                                         seq_idx=None, dt_limit=(0.0, float("inf")), return_final_states=False,
                                         activation="silu", rmsnorm_weight=None, rmsnorm_eps=1e-6, outproj_weight=None,
                                         outproj_bias=None, headdim=None, ngroups=1, norm_before_gate=True,
                                         dtype=torch.float64):
    """Same argument order as the call at block/mamba2.py:392-410.  zxbcdt: (B, L, 2*dim + 2*G*N + H)."""
    assert ngroups == 1 and initial_states is None and seq_idx is None and not return_final_states
    out_dtype = zxbcdt.dtype
    zx = zxbcdt.to(dtype)
    H = D.shape[0]
    dim = H * headdim
    N = (zx.shape[-1] - 2 * dim - H) // 2
    z, xBC, dt = zx[..., :dim], zx[..., dim:dim + dim + 2 * N], zx[..., -H:]
    xBC = causal_conv1d_ref(xBC.transpose(1, 2), conv1d_weight.reshape(conv1d_weight.shape[0], -1), conv1d_bias,
                            activation=activation, dtype=dtype).transpose(1, 2)
    x, Bm, Cm = xBC[..., :dim], xBC[..., dim:dim + N], xBC[..., dim + N:]
    dt = softplus_ref(dt + dt_bias.to(dtype))
    if dt_limit != (0.0, float("inf")):
        dt = dt.clamp(*dt_limit)
    y = ssd_scan_ref(x, dt, A, Bm, Cm, D, headdim, dtype=dtype)
    if rmsnorm_weight is not None:
        y = rmsnorm_gated_ref(y, z, rmsnorm_weight, rmsnorm_eps, norm_before_gate, dtype=dtype)
    else:
        y = y * (z * torch.sigmoid(z))
    if outproj_weight is not None:
        y = y @ outproj_weight.to(dtype).t()
        if outproj_bias is not None:
            y = y + outproj_bias.to(dtype)
    return y.to(out_dtype)


def mamba2_spiral_forward_ref(u, params, lists, headdim=64, eps=1e-5, dtype=torch.float64):
    """Mamba2.forward(u, 'spiral') of block/mamba2.py:359-457 as a pure function on reference-format params:
    in_proj.weight, conv1d.weight, conv1d.bias, dt_bias, A_log, D, norm.weight, out_proj.weight."""
    out_dtype = u.dtype
    order, order_rev, orig, orig_rev = (torch.as_tensor(t, dtype=torch.long) for t in lists)
    zx = u.to(dtype) @ params["in_proj.weight"].to(dtype).t()                   # (B, L, d_in_proj)
    A = -torch.exp(params["A_log"].to(dtype))
    outs = []
    for idx in (None, order, order_rev):                                        # CrossScan permutes dim 1 (block/mamba2.py:25-44)
        zk = zx if idx is None else zx[:, idx, :]
        outs.append(mamba_split_conv1d_scan_combined_ref(
            zk, params["conv1d.weight"], params["conv1d.bias"], params["dt_bias"], A, params["D"], chunk_size=256,
            activation="silu", rmsnorm_weight=params["norm.weight"], rmsnorm_eps=eps, outproj_weight=params["out_proj.weight"],
            outproj_bias=None, headdim=headdim, ngroups=1, norm_before_gate=False, dtype=dtype))
    out = outs[0] + outs[1][:, orig, :] + outs[2][:, orig_rev, :]               # CrossMerge
    return out.to(out_dtype)


def mamba2_baseline_forward_ref(u, params, scan_type, lists=None, headdim=64, eps=1e-5, dtype=torch.float64):
    """Mamba2.forward(u, scan_type) for 'zigma' | 'vim' | 'vmamba' (block/mamba2.py:459-615): the token axis (dim 1 of the
    (B, L, d_in_proj) in_proj output) is permuted per direction, one combined operator call per direction, outputs scattered
    back (and averaged for ViM, whose second output IS flipped along the token axis here, block/mamba2.py:502,522)."""
    out_dtype = u.dtype
    zx = u.to(dtype) @ params["in_proj.weight"].to(dtype).t()
    A = -torch.exp(params["A_log"].to(dtype))
    op = lambda zk: mamba_split_conv1d_scan_combined_ref(
        zk, params["conv1d.weight"], params["conv1d.bias"], params["dt_bias"], A, params["D"], chunk_size=256, activation="silu",
        rmsnorm_weight=params["norm.weight"], rmsnorm_eps=eps, outproj_weight=params["out_proj.weight"], outproj_bias=None,
        headdim=headdim, ngroups=1, norm_before_gate=False, dtype=dtype)
    lt = lambda v: torch.as_tensor(v, dtype=torch.long)
    if scan_type == "zigma":
        order, inv = lists
        out = op(zx[:, lt(order), :])[:, lt(inv), :]
    elif scan_type == "vim":
        out = (op(zx) + torch.flip(op(torch.flip(zx, [1])), [1])) / 2
    elif scan_type == "vmamba":
        orders, invs = lists
        out = sum(op(zx[:, lt(orders[k]), :])[:, lt(invs[k]), :] for k in range(4))
    else:
        raise ValueError(scan_type)
    return out.to(out_dtype)
