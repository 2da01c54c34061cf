"""fp64/fp32 CPU restatement of the Mamba-1 operator used by DiffMa.  TEST INFRASTRUCTURE ONLY.

The reference calls `mamba_inner_fn` from the un-vendored wheel mamba-ssm==2.0.4 (block/mamba.py:11, call sites :346-348).
This file restates that package's published reference recurrences (`selective_scan_ref`, `causal_conv1d_ref`, and the
composition in `mamba_inner_ref`; mathematics in SURVEY.md A.1) with plain sequential loops -- deliberately different in
structure from the HIP kernels.

PARITY PINNED (round 2): the reference itself holds this arithmetic in pure PyTorch -- `Mamba.step()`, block/mamba.py:405-448
(conv step, x_proj, dt_proj, softplus, exp(dt*A) state update, C.h, D skip, SiLU(z) gate, out_proj), reached when the absent
wheels' decode kernels are None.  tools/gen_golden.py runs it token by token in fp64 (G10, tests/golden/g10_reference_step.npz;
no function of this file takes part) and tests/test_golden_cpu.py::test_oracle_mamba_inner_matches_reference_step holds
`mamba_inner_ref` / `selective_scan_ref` (incl. the final state) to it at <= 1e-9.

Layout convention here is the REFERENCE's: channel-major (B, D, L) like block/mamba.py:333-337.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def softplus_ref(x: torch.Tensor) -> torch.Tensor:
    # upstream: x > 20 ? x : log1p(exp(x))   
    return torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False, dtype=torch.float64):
    """Sequential restatement of selective_scan_fn.

    u, delta, z: (B, D, L);  A: (D, N);  B, C: (B, N, L) or (B, G, N, L);  D, delta_bias: (D,).
    h_l = exp(delta_l A) h_{l-1} + delta_l B_l u_l ;  y_l = C_l.h_l + D u_l ;  out = y * silu(z).
    """
    out_dtype = u.dtype
    u = u.to(dtype)
    delta = delta.to(dtype)
    if delta_bias is not None:
        delta = delta + delta_bias.to(dtype)[None, :, None]
    if delta_softplus:
        delta = softplus_ref(delta)
    A = A.to(dtype)
    Bm, Cm = B.to(dtype), C.to(dtype)
    bsz, dim, L = u.shape
    N = A.shape[1]
    if Bm.dim() == 3:
        Bm = Bm[:, None]
    if Cm.dim() == 3:
        Cm = Cm[:, None]
    G = Bm.shape[1]
    rep = dim // G
    Bm = Bm.repeat_interleave(rep, dim=1)  # (B, D, N, L)
    Cm = Cm.repeat_interleave(rep, dim=1)
    h = torch.zeros(bsz, dim, N, dtype=dtype, device=u.device)
    ys = []
    for l in range(L):
        dl = delta[:, :, l]                                  # (B, D)
        a = torch.exp(dl[:, :, None] * A[None])              # (B, D, N)
        b = dl[:, :, None] * Bm[:, :, :, l] * u[:, :, l, None]
        h = a * h + b
        ys.append((h * Cm[:, :, :, l]).sum(-1))
    y = torch.stack(ys, dim=2)                               # (B, D, L)
    if D is not None:
        y = y + u * D.to(dtype)[None, :, None]
    if z is not None:
        zz = z.to(dtype)
        y = y * (zz * torch.sigmoid(zz))
    y = y.to(out_dtype)
    return (y, h) if return_last_state else y


def causal_conv1d_ref(x, weight, bias=None, activation=None, dtype=torch.float64):
    """x: (B, D, L); weight: (D, W); left zero padding of W-1; optional SiLU."""
    out_dtype = x.dtype
    x = x.to(dtype)
    w = weight.to(dtype)
    dim, W = w.shape
    L = x.shape[-1]
    y = torch.zeros_like(x)
    for j in range(W):
        shift = W - 1 - j            # tap j multiplies x[l - shift]
        if shift == 0:
            y = y + w[None, :, j, None] * x
        elif shift < L:
            y[:, :, shift:] = y[:, :, shift:] + w[None, :, j, None] * x[:, :, : L - shift]
    if bias is not None:
        y = y + bias.to(dtype)[None, :, None]
    if activation in ("silu", "swish"):
        y = y * torch.sigmoid(y)
    elif activation is not None:
        raise ValueError(activation)
    return y.to(out_dtype)


def mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                    out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, delta_softplus=True,
                    dtype=torch.float64, return_pre_proj=False):
    """Restatement of mamba_inner_fn with the exact argument order of block/mamba.py:346.

    xz: (B, 2*Din, L) channel-major.  Returns (B, L, d_model).
    """
    out_dtype = xz.dtype
    xz = xz.to(dtype)
    Din = xz.shape[1] // 2
    L = xz.shape[-1]
    x, z = xz[:, :Din], xz[:, Din:]
    w = conv1d_weight.reshape(Din, -1)
    xc = causal_conv1d_ref(x, w, conv1d_bias, activation="silu", dtype=dtype)
    Wx = x_proj_weight.to(dtype)
    Wdt = delta_proj_weight.to(dtype)
    R = Wdt.shape[1]
    N = A.shape[1]
    x_dbl = torch.einsum("bdl,ed->ble", xc, Wx)              # (B, L, R+2N)
    delta = torch.einsum("blr,dr->bdl", x_dbl[..., :R], Wdt)  # (B, Din, L)
    assert B is None and C is None, "DiffMa always uses input-dependent B and C"
    Bm = x_dbl[..., R:R + N].permute(0, 2, 1)                # (B, N, L)
    Cm = x_dbl[..., R + N:R + 2 * N].permute(0, 2, 1)
    y = selective_scan_ref(xc, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias,
                           delta_softplus=delta_softplus, dtype=dtype)
    if return_pre_proj:
        return y.to(out_dtype)                               # (B, Din, L)
    out = torch.einsum("bdl,ed->ble", y, out_proj_weight.to(dtype))
    if out_proj_bias is not None:
        out = out + out_proj_bias.to(dtype)
    return out.to(out_dtype)


# ---- token reindex (block/mamba.py:26-82) ------------------------------------------------------------
def cross_scan_ref(x, order, order_reversal):
    """CrossScan.forward: xs[:,0]=x, xs[:,1]=x[:,:,order], xs[:,2]=x[:,:,order_reversal]; x (B, C, L)."""
    idx1 = torch.as_tensor(order, dtype=torch.long)
    idx2 = torch.as_tensor(order_reversal, dtype=torch.long)
    return torch.stack([x, x[:, :, idx1], x[:, :, idx2]], dim=1)


def cross_merge_ref(ys, origina, origina_reversal):
    """CrossMerge.forward: y = ys[:,0] + ys[:,1][:, origina, :] + ys[:,2][:, origina_rev, :]; ys (B,3,L,C)."""
    i1 = torch.as_tensor(origina, dtype=torch.long)
    i2 = torch.as_tensor(origina_reversal, dtype=torch.long)
    return ys[:, 0] + ys[:, 1][:, i1, :] + ys[:, 2][:, i2, :]


def mamba_spiral_forward_ref(hidden, params, lists, dtype=torch.float64):
    """Mamba.forward(hidden, 'spiral') of block/mamba.py:317-355 as a pure function.

    params: dict with in_proj.weight, conv1d.weight, conv1d.bias, x_proj.weight, dt_proj.weight,
            dt_proj.bias, A_log, D, out_proj.weight      (names = the reference's state-dict keys)
    lists:  (token_list, token_list_reversal, origina_list, origina_list_reversal)
    """
    out_dtype = hidden.dtype
    hs = hidden.to(dtype)
    Wi = params["in_proj.weight"].to(dtype)
    xz = torch.einsum("ed,bld->bel", Wi, hs)                 # (B, 2Din, L)  block/mamba.py:333-337
    A = -torch.exp(params["A_log"].to(torch.float64 if dtype == torch.float64 else torch.float32))
    order, order_rev, orig, orig_rev = lists
    xs = cross_scan_ref(xz, order, order_rev)
    outs = []
    for k in range(3):
        outs.append(mamba_inner_ref(
            xs[:, k], params["conv1d.weight"], params["conv1d.bias"], params["x_proj.weight"],
            params["dt_proj.weight"], params["out_proj.weight"], None, A, None, None,
            params["D"], delta_bias=params["dt_proj.bias"], delta_softplus=True, dtype=dtype))
    out = cross_merge_ref(torch.stack(outs, dim=1), orig, orig_rev)
    return out.to(out_dtype)


# ---- baseline scan orders (block/mamba.py:85-224 reindexing, 357-401 call pattern) ---------------------------------
def zig_lists_ref(n, i):
    """Independent restatement of tools.zig (reference tools.py:46-128): boustrophedon numbering of the n x n grid, by rows
    (odd variants) or columns (even), mirrored left-right (3, 4, 7, 8) and/or top-bottom (5..8).  Returns (order, inverse)."""
    v = i % 8 or 8
    order = []
    for r in range(n):
        for c in range(n):
            rr = n - 1 - r if v >= 5 else r
            cc = n - 1 - c if v in (3, 4, 7, 8) else c
            if v % 2 == 1:
                order.append(rr * n + (cc if rr % 2 == 0 else n - 1 - cc))
            else:
                order.append(cc * n + (rr if cc % 2 == 0 else n - 1 - rr))
    inv = [0] * (n * n)
    for pos, tok in enumerate(order):
        inv[tok] = pos
    return order, inv


def vmamba_lists_ref(n):
    """tools.vmamba_ (reference tools.py:130-152): zigzag variants 1, 2, 7, 8."""
    pairs = [zig_lists_ref(n, v) for v in (1, 2, 7, 8)]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def mamba_baseline_forward_ref(hidden, params, scan_type, lists=None, dtype=torch.float64):
    """Mamba.forward(hidden, scan_type) for scan_type in 'zigma' | 'vim' | 'vmamba' | 'eff' (block/mamba.py:357-401) as a
    pure function; one mamba_inner_ref (conv .. out_proj) per direction, combined exactly as the reference combines them.
    lists: (order, inverse) for 'zigma', (orders[4], inverses[4]) for 'vmamba', None otherwise."""
    out_dtype = hidden.dtype
    hs = hidden.to(dtype)
    xz = torch.einsum("ed,bld->bel", params["in_proj.weight"].to(dtype), hs)          # (B, 2Din, L)
    A = -torch.exp(params["A_log"].to(torch.float64 if dtype == torch.float64 else torch.float32))
    inner = lambda t: mamba_inner_ref(t, params["conv1d.weight"], params["conv1d.bias"], params["x_proj.weight"],
                                      params["dt_proj.weight"], params["out_proj.weight"], None, A, None, None, params["D"],
                                      delta_bias=params["dt_proj.bias"], delta_softplus=True, dtype=dtype)
    lt = lambda v: torch.as_tensor(v, dtype=torch.long)
    if scan_type == "zigma":                                   # gather by `order`, scatter back by its inverse (:357-360)
        order, inv = lists
        out = inner(xz[:, :, lt(order)])[:, lt(inv), :]
    elif scan_type == "vim":                                   # (:362-367) the second output is flipped along dim 2 of a
        out1 = inner(xz)                                       # (B, L, D) tensor, i.e. along the FEATURE axis
        out2 = inner(torch.flip(xz, [2]))
        out = (out1 + torch.flip(out2, [2])) / 2
    elif scan_type == "vmamba":                                # (:369-382) four gathers, four scatters, sum
        orders, invs = lists
        out = sum(inner(xz[:, :, lt(orders[k])])[:, lt(invs[k]), :] for k in range(4))
    elif scan_type == "eff":                                   # (:384-399) 2x2 atrous split of the token grid
        Bsz, _, L = xz.shape
        n = int(round(L ** 0.5))
        grid = xz.reshape(Bsz, -1, n, n)                       # [.., row, col]
        gt = grid.transpose(2, 3)                              # [.., col, row]
        subs = [grid[:, :, ::2, ::2], gt[:, :, ::2, 1::2], grid[:, :, ::2, 1::2], gt[:, :, 1::2, 1::2]]
        outs = [inner(sb.reshape(Bsz, xz.shape[1], -1)) for sb in subs]           # each (B, L/4, d_model)
        h = n // 2
        full = outs[0].new_empty(Bsz, n, n, outs[0].shape[-1])
        full[:, ::2, ::2] = outs[0].reshape(Bsz, h, h, -1)
        full[:, 1::2, ::2] = outs[1].reshape(Bsz, h, h, -1).transpose(1, 2)
        full[:, ::2, 1::2] = outs[2].reshape(Bsz, h, h, -1)
        full[:, 1::2, 1::2] = outs[3].reshape(Bsz, h, h, -1).transpose(1, 2)
        out = full.reshape(Bsz, L, -1)
    else:
        raise ValueError(scan_type)
    return out.to(out_dtype)
