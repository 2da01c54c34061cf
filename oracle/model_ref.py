"""Functional CPU restatement of the DiffMa denoiser forward.  TEST INFRASTRUCTURE ONLY.

Follows reference model.py:264-301 (DiffMa.forward), model.py:22-109 (PatchEmbed, TimestepEmbed, FinalLayer),
block/mamba_block.py:100-115 (Spiral_MambaBlock.forward) and block/mamba.py:317-355 (Mamba.forward,
'spiral'), operating directly on a reference-format state dict.  Pinned by tests/golden/g5_tiny_diffma.npz
(the output of the reference's own classes with the oracle operator stubbed in).  Also the timed
"port" CPU baseline of bench.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from .mamba2_ref import mamba2_baseline_forward_ref, mamba2_spiral_forward_ref
from .mamba_ref import mamba_baseline_forward_ref, mamba_spiral_forward_ref, vmamba_lists_ref, zig_lists_ref


def spiral_lists_ref(n):
    """Independent numpy restatement of tools.spiral (reference tools.py:2-43): coordinates of an
    outward square spiral are generated ring by ring from run lengths, then ranked."""
    dir_sets = [((0, 1), (1, 0), (0, -1), (-1, 0)), ((1, 0), (0, -1), (-1, 0), (0, 1)), ((0, -1), (-1, 0), (0, 1), (1, 0)),
                ((-1, 0), (0, 1), (1, 0), (0, -1)), ((0, 1), (-1, 0), (0, -1), (1, 0)), ((0, -1), (1, 0), (0, 1), (-1, 0)),
                ((1, 0), (0, 1), (-1, 0), (0, -1)), ((-1, 0), (0, -1), (1, 0), (0, 1))]
    orders = []
    for dirs in dir_sets:
        # unbounded walk long enough to cover the grid from the (possibly off-centre) start
        runs = np.repeat(np.arange(1, 2 * n + 3), 2)                       # 1,1,2,2,3,3,...
        step_dir = np.repeat(np.arange(len(runs)) % 4, runs)               # direction index of every unit step
        d = np.asarray(dirs)[step_dir]                                     # [steps, 2]
        pos = np.concatenate([[[n // 2, n // 2]], n // 2 + np.cumsum(d, axis=0)])   # position BEFORE each step + last
        pos = pos[:-1]
        inside = (pos[:, 0] >= 0) & (pos[:, 0] < n) & (pos[:, 1] >= 0) & (pos[:, 1] < n)
        cells = pos[inside][: n * n]
        rank = np.empty(n * n, dtype=np.int64)
        rank[cells[:, 0] * n + cells[:, 1]] = np.arange(n * n)
        orders.append(rank)
        orders.append(n * n - 1 - rank)
    orders = np.stack(orders)
    inv = np.empty_like(orders)
    for k in range(orders.shape[0]):
        inv[k, orders[k]] = np.arange(n * n)
    return orders, inv


def _ln(x, weight=None, bias=None, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def _timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def diffma_forward_ref(sd, x, t, y, y2, w, *, patch_size, depth, dtype=torch.float32, return_blocks=False, use_mamba2=False,
                       headdim=64, block_type="spiral"):
    """sd: reference-format state dict (tensors).  Inputs as DiffMa.forward (model.py:264).
    block_type: 'spiral' (DiffMa) or one of the baseline blocks 'zig' | 'vim' | 'vmamba' | 'efficientVMamba'
    (block/mamba_block.py:190-198, 247-255, 318-326, 381-389; pinned by tests/golden/g9_baseline_blocks.npz)."""
    g = lambda k: sd[k].to(dtype)
    x = x.to(dtype)
    p = patch_size
    h = F.conv2d(x, g("x_embedder.proj.weight"), g("x_embedder.proj.bias"), stride=p)
    h = h.flatten(2).transpose(1, 2) + g("pos_embed")
    n_side = int(round(math.sqrt(h.shape[1])))
    orders, inverses = spiral_lists_ref(n_side)
    temb = _timestep_embedding(t, g("t_embedder.mlp.0.weight").shape[1]).to(dtype)
    temb = F.linear(F.silu(F.linear(temb, g("t_embedder.mlp.0.weight"), g("t_embedder.mlp.0.bias"))),
                    g("t_embedder.mlp.2.weight"), g("t_embedder.mlp.2.bias"))
    c = torch.cat([temb + y.to(dtype), temb + y2.to(dtype).mean(dim=1)], dim=1)
    w = w.to(dtype)
    outs = []
    for i in range(depth):
        if i == 0:
            inp = h
        elif i > depth / 2:
            inp = outs[-1] + outs[depth - i - 1]
        else:
            inp = outs[-1]
        pre = f"blocks.{i}."
        k = (2 * i) % 16
        lists = (orders[k].tolist(), orders[k + 1].tolist(), inverses[k].tolist(), inverses[k + 1].tolist())
        mod = F.linear(F.silu(c), g(pre + "adaLN_modulation.1.weight"), g(pre + "adaLN_modulation.1.bias"))
        shift, scale, gate = mod.chunk(3, dim=1)
        xs = _ln(inp, g(pre + "norm1.weight"), g(pre + "norm1.bias")) * (1 + scale[:, None]) + shift[:, None]
        sub = lambda name: {kk[len(pre + name) + 1:]: v for kk, v in sd.items() if kk.startswith(pre + name + ".")}
        if block_type != "spiral":                       # one mixer, no soft mask, no fusion
            scan_type = {"zig": "zigma", "vim": "vim", "vmamba": "vmamba", "efficientVMamba": "eff"}[block_type]
            bl = zig_lists_ref(n_side, i) if block_type == "zig" else (vmamba_lists_ref(n_side) if block_type == "vmamba" else None)
            if use_mamba2:
                h = inp + gate[:, None] * mamba2_baseline_forward_ref(xs, sub("mamba"), scan_type, bl, headdim=headdim, dtype=dtype)
            else:
                h = inp + gate[:, None] * mamba_baseline_forward_ref(xs, sub("mamba"), scan_type, bl, dtype=dtype)
            outs.append(h)
            continue
        ws = xs * w
        if use_mamba2:
            mix = lambda name, inp_: mamba2_spiral_forward_ref(inp_, sub(name), lists, headdim=headdim, dtype=dtype)
        else:
            mix = lambda name, inp_: mamba_spiral_forward_ref(inp_, sub(name), lists, dtype=dtype)
        xs, ws = mix("mamba1", xs), mix("mamba2", ws)
        cat = torch.cat([xs, ws], dim=-1)
        a = _ln(cat, g(pre + "attention_network.0.weight"), g(pre + "attention_network.0.bias"))
        a = F.linear(F.silu(F.linear(a, g(pre + "attention_network.1.weight"), g(pre + "attention_network.1.bias"))),
                     g(pre + "attention_network.3.weight"), g(pre + "attention_network.3.bias"))
        a = torch.sigmoid(a)
        h = inp + gate[:, None] * (a * xs + (1 - a) * ws)
        outs.append(h)
    mod = F.linear(F.silu(c), g("final_layer.adaLN_modulation.1.weight"), g("final_layer.adaLN_modulation.1.bias"))
    shift, scale = mod.chunk(2, dim=1)
    o = _ln(h, eps=1e-6) * (1 + scale[:, None]) + shift[:, None]
    o = F.linear(o, g("final_layer.linear.weight"), g("final_layer.linear.bias"))
    cch = o.shape[-1] // (p * p)
    o = o.reshape(o.shape[0], n_side, n_side, p, p, cch).permute(0, 5, 1, 3, 2, 4).reshape(o.shape[0], cch, n_side * p, n_side * p)
    return (o, outs) if return_blocks else o
