"""DiffMa denoiser and the DiffMa_models factory (reference model.py:22-316, 377-420, 634-640).

`DiffMa_models[name](input_size=, dt_rank=, d_state=, use_mamba2=)` and `model(x, t, y=, y2=, w=)` keep the
reference's call contract (train.py:130-135, sample.py:42-46, model.py:264); parameter names match the
reference state dict (pos_embed, x_embedder.proj.*, t_embedder.mlp.{0,2}.*, blocks.N.*, final_layer.*).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from .mamba_block import (EfficientVMamba_MambaBlock, Spiral_MambaBlock, ViM_MambaBlock, VMamba_MambaBlock, Zig_MambaBlock,
                          modulate)
from .tools import spiral, vmamba_, zig


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    """Strided-conv patchify: (B, C, H, W) -> (B, T, D)."""

    def __init__(self, img_size=28, patch_size=2, stride=2, in_chans=4, embed_dim=512, norm_layer=None, flatten=True):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.grid_size = tuple((self.img_size[i] - self.patch_size[i]) // stride + 1 for i in (0, 1))
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=stride)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        H, W = x.shape[-2:]
        assert (H, W) == self.img_size, f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        ph, pw = self.patch_size
        if self.proj.stride == (ph, pw) and self.proj.padding == (0, 0):
            # stride == kernel: the conv is a GEMM over non-overlapping patches.  Doing it as one (MFMA) matmul keeps
            # both passes on hipBLASLt; MIOpen's bf16 weight-gradient path for this shape is a naive kernel (2 ms/call).
            gh, gw = self.grid_size
            B, C = x.shape[:2]
            xp = x[:, :, :gh * ph, :gw * pw].reshape(B, C, gh, ph, gw, pw).permute(0, 2, 4, 1, 3, 5).reshape(B, gh * gw, C * ph * pw)
            x = torch.nn.functional.linear(xp, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
            if not self.flatten:
                x = x.transpose(1, 2).reshape(B, -1, gh, gw)
        else:
            x = self.proj(x)
            if self.flatten:
                x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class TimestepEmbed(nn.Module):
    """Sinusoidal features -> 2-layer MLP (reference model.py:49-85)."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, t):
        return self.mlp(self.timestep_embedding(t, self.frequency_embedding_size))


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(2 * hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
        return self.linear(modulate(self.norm_final(x), shift, scale))


class DiffMa(nn.Module):
    def __init__(self, input_size=28, patch_size=2, strip_size=2, in_channels=4, hidden_size=512, depth=16,
                 learn_sigma=True, block_type="spiral", dt_rank=16, d_state=16, use_mamba2=False):
        super().__init__()
        if block_type not in ("spiral", "zig", "vim", "vmamba", "efficientVMamba"):
            raise NotImplementedError(f"block_type={block_type!r}: the Mamba blocks are built (spiral, zig, vim, vmamba, "
                                      "efficientVMamba); the attention baseline 'DiT' is not on the scan path (SURVEY.md 8f)")
        self.learn_sigma, self.depth, self.in_channels = learn_sigma, depth, in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size, self.input_size, self.block_type = patch_size, input_size, block_type
        self.x_embedder = PatchEmbed(input_size, patch_size, strip_size, in_channels, hidden_size)
        self.t_embedder = TimestepEmbed(hidden_size)
        num_patches = self.x_embedder.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, hidden_size), requires_grad=False)
        side = int(input_size / patch_size)
        common = dict(D_dim=hidden_size, E_dim=2 * hidden_size, dim_inner=2 * hidden_size, dt_rank=dt_rank, d_state=d_state,
                      use_mamba2=use_mamba2)
        if block_type == "spiral":
            orders, inverses = spiral(side)
            nlist = len(orders)
            blocks = [Spiral_MambaBlock(token_list=orders[(2 * i) % nlist], token_list_reversal=orders[(2 * i) % nlist + 1],
                                        origina_list=inverses[(2 * i) % nlist], origina_list_reversal=inverses[(2 * i) % nlist + 1],
                                        **common) for i in range(depth)]
        elif block_type == "zig":             # block i scans in zigzag variant i % 8 (reference model.py:159-170)
            blocks = [Zig_MambaBlock(token_list=zig(side, i)[0], origina_list=zig(side, i)[1], **common) for i in range(depth)]
        elif block_type == "vim":             # forward + backward raster scans (model.py:171-180)
            blocks = [ViM_MambaBlock(**common) for _ in range(depth)]
        elif block_type == "vmamba":          # the same four scans in every block (model.py:181-193)
            order_list, original_list = vmamba_(side)
            blocks = [VMamba_MambaBlock(token_list=order_list, origina_list=original_list, **common) for _ in range(depth)]
        else:                                 # 'efficientVMamba': four atrous sub-grid scans (model.py:194-203)
            blocks = [EfficientVMamba_MambaBlock(**common) for _ in range(depth)]
        self.blocks = nn.ModuleList(blocks)
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for blk in self.blocks:                       # adaLN-zero
            nn.init.constant_(blk.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(blk.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def unpatchify(self, x):
        c, p = self.out_channels, self.x_embedder.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c).permute(0, 5, 1, 3, 2, 4)      # n c h p w q
        return x.reshape(x.shape[0], c, h * p, w * p)

    def forward(self, x, t, y, y2, w):
        """x (N,C,H,W) latents; t (N,) timesteps; y (N,D) CLIP embedding; y2 (N,T,D) CT tokens; w (N,T,1) soft mask."""
        if x.is_cuda and torch.is_grad_enabled():
            from . import step_prep
            step_prep.prepare(self)                    # A = -exp(A_log) of every mixer and the 16-bit weight copies: a few foreach launches
        x = self.x_embedder(x) + self.pos_embed
        t = self.t_embedder(t)
        c = torch.cat((t + y, t + y2.mean(dim=1)), dim=1)
        mods = None
        if x.is_cuda and torch.is_grad_enabled() and self.block_type == "spiral":
            from .mamba_block import adaln_all
            mods = adaln_all(self, c)                  # every block's (shift, scale, gate) from ONE product over the stacked adaLN weights
        kw = (lambda i: {"mod": mods[i]}) if mods is not None else (lambda i: {})
        outs = []
        for i, blk in enumerate(self.blocks):          # U-ViT style long skips (reference model.py:286-295)
            if i == 0:
                x = blk(x, c, w, **kw(i))
            elif i > self.depth / 2:
                x = blk(outs[-1] + outs[self.depth - i - 1], c, w, **kw(i))
            else:
                x = blk(outs[-1], c, w, **kw(i))
            outs.append(x)
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            from .block_ops import drop_mask_cache
            drop_mask_cache()                          # the blocks' shared mask copy belongs to this capture's pool: not to the next one
        return self.unpatchify(self.final_layer(x, c))

    def forward_with_cfg(self, x, t, y, y2, w, cfg_scale):
        half = x[: len(x) // 2]
        out = self.forward(torch.cat([half, half], dim=0), t, y, y2, w)
        eps, rest = out[:, :3], out[:, 3:]
        cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
        g = uncond + cfg_scale * (cond - uncond)
        return torch.cat([torch.cat([g, g], dim=0), rest], dim=1)


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.outer(pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed_from_grid(embed_dim, grid):
    assert embed_dim % 2 == 0
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                           get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)   # w first (MAE convention)
    pe = get_2d_sincos_pos_embed_from_grid(embed_dim, grid)
    if cls_token and extra_tokens > 0:
        pe = np.concatenate([np.zeros([extra_tokens, embed_dim]), pe], axis=0)
    return pe


_DEPTH = {"S": 4, "B": 8, "L": 16, "XL": 28, "XXL": 56}


def _make(depth, patch, block_type="spiral"):
    def ctor(**kwargs):
        return DiffMa(depth=depth, hidden_size=512, patch_size=patch, strip_size=patch, block_type=block_type, **kwargs)
    return ctor


# 'DiffMa-{S,B,L,XL,XXL}/{2,4,7}'  (reference model.py:636-640)
DiffMa_models = {f"DiffMa-{size}/{patch}": _make(depth, patch) for size, depth in _DEPTH.items() for patch in (2, 4, 7)}
# the baseline families the reference reproduces on the same mixer: '{ZigMa,ViM,VMamba,EMamba}-{S,B,L,XL}/{2,4,7}' and
# '-BL/2' (depth 13)  (reference model.py:641-664)
for _family, _bt in (("ZigMa", "zig"), ("ViM", "vim"), ("VMamba", "vmamba"), ("EMamba", "efficientVMamba")):
    for _size in ("S", "B", "L", "XL"):
        for _patch in (2, 4, 7):
            DiffMa_models[f"{_family}-{_size}/{_patch}"] = _make(_DEPTH[_size], _patch, _bt)
    DiffMa_models[f"{_family}-BL/2"] = _make(13, 2, _bt)
