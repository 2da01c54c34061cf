"""CT token embedder that produces the soft mask `w` and the token conditioning `y2` of DiffMa.forward
(reference block/CT_encoder.py:5-45 + block/visionEmbedding.py:4-73; frozen in train.py:158-169 / sample.py:60-69).

    tokens = patch-embed conv(x)                                   [B, T, E]      T = (img/patch)^2
    w      = sigmoid( fc(mean_E tokens) + fc(max_E tokens) )       [B, T, 1]      fc: T -> T/14 -> T (ReLU)
    y2     = LayerNorm_E(tokens * w)                               [B, T, E]

State-dict keys match the reference module (9 tensors: vision_embedding.proj.{weight,bias},
vision_embedding.mask_token, fc.{0,2}.{weight,bias}, norm.{weight,bias}), so `pretrain_ct_encoder/patch_size_*.pt`
loads unchanged.  This is a 263 KB frozen pre-processing net (<0.1 % of a step): plain PyTorch, it runs wherever its
input lives.  The reference's adaptive (T, 1) poolings over a [B, T, E] tensor are a mean / max over E.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class VisionEmbedding(nn.Module):
    """Patch embedding with optional mask / cls tokens (block/visionEmbedding.py)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, contain_mask_token=False, prepend_cls_token=False):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.patch_shape = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.patch_shape[0] * self.patch_shape[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if contain_mask_token else None
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if prepend_cls_token else None

    def num_position_embeddings(self):
        return self.num_patches + (0 if self.cls_token is None else 1)

    def forward(self, x, masked_position=None, **kwargs):
        if tuple(x.shape[-2:]) != self.img_size:
            raise ValueError(f"input size {tuple(x.shape[-2:])} does not match the embedder's {self.img_size}")
        tok = self.proj(x).flatten(2).transpose(1, 2)                     # [B, T, E]
        if masked_position is not None:
            if self.mask_token is None:
                raise ValueError("masked_position needs contain_mask_token=True")
            m = masked_position.unsqueeze(-1).to(tok.dtype)
            tok = torch.lerp(tok, self.mask_token.to(tok.dtype).expand_as(tok), m)
        if self.cls_token is not None:
            tok = torch.cat((self.cls_token.expand(tok.shape[0], -1, -1), tok), dim=1)
        return tok


class CT_Encoder(nn.Module):
    def __init__(self, img_size=28, patch_size=2, in_channels=4, embed_dim=1024, contain_mask_token=True, reduction_ratio=14):
        super().__init__()
        self.vision_embedding = VisionEmbedding(img_size=img_size, patch_size=patch_size, in_chans=in_channels,
                                                embed_dim=embed_dim, contain_mask_token=contain_mask_token)
        tokens = int((img_size / patch_size) ** 2)
        self.fc = nn.Sequential(nn.Linear(tokens, int(tokens / reduction_ratio)), nn.ReLU(inplace=True),
                                nn.Linear(int(tokens / reduction_ratio), tokens))
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        """x: [B, C, H, W] latent -> (w [B, T, 1], y2 [B, T, E])."""
        tok = self.vision_embedding(x)
        logits = self.fc(tok.amax(dim=-1)) + self.fc(tok.mean(dim=-1))
        w = torch.sigmoid(logits).unsqueeze(-1)
        return w, F.layer_norm(tok * w, (tok.shape[-1],), self.norm.weight, self.norm.bias, self.norm.eps)
