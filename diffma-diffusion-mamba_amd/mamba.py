"""Mamba-1 mixer of DiffMa, 'spiral' scan (reference block/mamba.py:226-355), MI355X-native; the scan orders of the
baseline blocks ('zigma', 'vim', 'vmamba', 'eff': block/mamba.py:85-224, 357-401) run on the same fused operator.

State-dict keys and constructor arguments match the reference class so its checkpoints load:
    in_proj.weight (2*Din, d_model)   conv1d.weight (Din, 1, d_conv)   conv1d.bias (Din)
    x_proj.weight (R+2N, Din)         dt_proj.weight (Din, R)          dt_proj.bias (Din)
    A_log (Din, N)   D (Din)          out_proj.weight (d_model, Din)
Differences in HOW (not WHAT): activations stay token-major end to end, the three scan directions are one
fused operator (selective_scan_interface.spiral_ssm), the permutations are int32 device buffers built
once (the reference re-uploads Python lists on every call, block/mamba.py:27,30), and the merge happens
before out_proj so there is one projection GEMM instead of three.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .selective_scan_interface import linear_pair, linear_splitk, spiral_ssm, spiral_ssm_pair, spiral_ssm_pair_supported
from .tools import efficient_scan_tokens


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, token_list=(),
                 token_list_reversal=(), origina_list=(), origina_list_reversal=()):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.layer_idx = layer_idx
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, kernel_size=d_conv, groups=self.d_inner,
                                padding=d_conv - 1, bias=conv_bias, **fk)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * d_state, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        # dt_proj init of the Mamba paper (block/mamba.py:284-302).  DiffMa's own initialize_weights later
        # overwrites it with xavier/zero; both are reproduced, in that order.
        std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -std, std)
        else:
            raise NotImplementedError(dt_init)
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = dt.clamp(min=dt_init_floor)
        with torch.no_grad():
            self.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))       # softplus^-1
        self.dt_proj.bias._no_reinit = True
        A = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1)
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

        self.token_list, self.token_list_reversal = list(token_list), list(token_list_reversal)
        self.origina_list, self.origina_list_reversal = list(origina_list), list(origina_list_reversal)
        # token_list is one permutation (spiral, zigma) or a list of four (vmamba, reference model.py:182-186)
        nested = bool(self.token_list) and isinstance(self.token_list[0], (list, tuple))
        L = len(self.token_list[0]) if nested else len(self.token_list)
        if L and not nested and self.token_list_reversal:
            idx = torch.tensor([list(range(L)), self.token_list, self.token_list_reversal], dtype=torch.int32)
        else:
            idx = torch.zeros((3, 0), dtype=torch.int32)
        # not in the state dict (the reference keeps Python lists)
        self.register_buffer("scan_index", idx, persistent=False)
        self._tables = {}                  # (scan_type, L, device) -> index tensors of the baseline scan orders
        self.vim_flip = "reference"        # 'reference': flip the feature axis of the backward output (block/mamba.py:366,
                                           # the token order stays reversed); 'token': the intended flip

    def _baseline_tables(self, scan_type, L, device):
        key = (scan_type, L, str(device))
        tab = self._tables.get(key)
        if tab is not None:
            return tab
        i32 = lambda rows: torch.tensor(rows, dtype=torch.int32, device=device)
        if scan_type == "zigma":                                   # one permutation per block (block/mamba.py:357-360)
            tab = (i32([self.token_list]), None)
        elif scan_type == "vmamba":                                # four permutations, outputs summed (block/mamba.py:369-382)
            tab = (i32(self.token_list), None)
        elif scan_type == "vim":                                   # forward + time-reversed scan (block/mamba.py:362-367)
            fwd, rev = list(range(L)), list(range(L - 1, -1, -1))
            tab = (i32([fwd, rev]), i32([fwd, fwd]) if self.vim_flip == "reference" else None)
        elif scan_type == "eff":                                   # four atrous sub-grids of L/4 tokens (block/mamba.py:384-399)
            n = int(round(math.sqrt(L)))
            if n * n != L:
                raise ValueError(f"EfficientVMamba needs a square token grid, got L={L}")
            tok = torch.from_numpy(efficient_scan_tokens(n)).reshape(-1)
            inv = torch.empty_like(tok)
            inv[tok] = torch.arange(L)
            tab = (i32([list(range(L // 4))]), (tok.to(device), inv.to(device)))
        else:
            raise NotImplementedError(f"scan_type={scan_type!r}")
        if tab[0].shape[1] != (L // 4 if scan_type == "eff" else L):
            raise ValueError(f"sequence length {L} != scan table length {tab[0].shape[1]}")
        self._tables[key] = tab
        return tab

    def forward(self, hidden_states, scan_type="spiral", inference_params=None):
        """hidden_states: (B, L, d_model) -> (B, L, d_model)."""
        if inference_params is not None:
            raise NotImplementedError("autoregressive decode is never used by a diffusion model")
        if scan_type != "spiral":
            return self._forward_baseline(hidden_states, scan_type)
        if hidden_states.shape[1] != self.scan_index.shape[1]:
            raise ValueError(f"sequence length {hidden_states.shape[1]} != spiral table length {self.scan_index.shape[1]}")
        xz = linear_splitk(hidden_states, self.in_proj.weight, self.in_proj.bias)      # [B, L, 2*Din] token-major
        A = self._A_now(xz)
        y = spiral_ssm(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                       self.dt_proj.bias, A, self.D, self.scan_index)
        return linear_splitk(y.to(xz.dtype), self.out_proj.weight, self.out_proj.bias)

    def _A_now(self, like):
        """A = -exp(A_log) for this call: the tensor step_prep made for all mixers at once, else computed / cached here."""
        A = self.__dict__.pop("_A_step", None)       # made for all mixers at once by step_prep.prepare (DiffMa.forward), used once
        if A is not None and A.device == like.device and torch.is_grad_enabled():
            return A
        if torch.is_grad_enabled() or (like.is_cuda and torch.cuda.is_current_stream_capturing()):
            return -torch.exp(self.A_log.float())    # inside a capture A is recomputed IN the graph: replays follow A_log
        # inference: A only changes when A_log does (two tiny kernels per call otherwise)
        cache = getattr(self, "_A_cache", None)      # (hipGraph replays of an optimizer do not bump _version: GraphedTrainStep drops the cache)
        if cache is None or cache[0] != self.A_log._version or cache[1].device != self.A_log.device:
            cache = (self.A_log._version, -torch.exp(self.A_log.detach().float()))
            self._A_cache = cache
        return cache[1]

    def _forward_baseline(self, hidden_states, scan_type):
        """The ZigMa / ViM / VMamba / EfficientVMamba token orders (reference block/mamba.py:357-401) on the fused operator.
        The reference runs one mamba_inner_fn (conv .. out_proj) per direction and combines the projected outputs; out_proj is
        linear and has no bias here, so the directions are summed before ONE projection wherever the reference sums them."""
        Bsz, L, _ = hidden_states.shape
        gather, extra = self._baseline_tables(scan_type, L, hidden_states.device)
        xz = linear_splitk(hidden_states, self.in_proj.weight, self.in_proj.bias)      # [B, L, 2*Din] token-major
        A = self.__dict__.pop("_A_step", None)
        if A is None or A.device != xz.device or not torch.is_grad_enabled():
            A = -torch.exp(self.A_log.float())
        ssm = lambda inp, **kw: spiral_ssm(inp, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                                           self.dt_proj.bias, A, self.D, gather, **kw)
        proj = lambda y: linear_splitk(y.to(xz.dtype), self.out_proj.weight, self.out_proj.bias)
        ndir = gather.shape[0]
        if self.out_proj.bias is not None and ndir > 1 and scan_type != "vim":
            raise NotImplementedError("out_proj bias with summed directions (the reference adds it once per direction)")
        if scan_type in ("zigma", "vmamba"):
            return proj(ssm(xz))
        if scan_type == "vim":
            if extra is None:                                      # vim_flip == 'token': average of the two scans, both in token order
                return proj(ssm(xz) * 0.5) if self.out_proj.bias is None else proj(ssm(xz)) * 0.5 + 0.5 * self.out_proj.bias
            ydir = ssm(xz, out_index=extra, merge=False)           # [2, B, L, Din]; direction 1 stays in reversed token order
            return (proj(ydir[0]) + torch.flip(proj(ydir[1]), [2])) / 2      # literal block/mamba.py:366-367
        # 'eff': every token belongs to exactly one of four L/4-token scans; gather them into a 4B batch and scatter back
        tok, inv = extra
        L4 = L // 4
        sub = xz[:, tok].view(Bsz, 4, L4, xz.shape[-1]).transpose(0, 1).reshape(4 * Bsz, L4, xz.shape[-1])
        out = proj(ssm(sub.contiguous()))                          # [4B, L/4, d_model]
        out = out.view(4, Bsz, L4, -1).transpose(0, 1).reshape(Bsz, L, -1)
        return out[:, inv]


def forward_pair(mix0: Mamba, mix1: Mamba, x0, x1):
    """(mix0(x0, 'spiral'), mix1(x1, 'spiral')) -- the two mixers of a DiffMa block (reference block/mamba_block.py:107-108) -- with
    every stage launched once for both when the call pattern allows it (selective_scan_interface._SpiralSSMPairFn), else one after
    the other.  Same arithmetic per mixer either way."""
    if not (isinstance(mix0, Mamba) and isinstance(mix1, Mamba) and x0.is_cuda and x0.shape == x1.shape and x0.dtype == x1.dtype
            and mix0.in_proj.bias is None and mix1.in_proj.bias is None and mix0.out_proj.bias is None and mix1.out_proj.bias is None
            and mix0.in_proj.weight.shape == mix1.in_proj.weight.shape and x0.shape[1] == mix0.scan_index.shape[1]):
        return mix0(x0, "spiral"), mix1(x1, "spiral")
    dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x0.dtype
    if not spiral_ssm_pair_supported(x0.shape[0], x0.shape[1], dt, mix0, mix1):
        return mix0(x0, "spiral"), mix1(x1, "spiral")
    xz0, xz1 = linear_pair(x0, x1, mix0.in_proj.weight, mix1.in_proj.weight)          # [B, L, 2*Din] each, halves of one buffer
    y0, y1 = spiral_ssm_pair(xz0, xz1, mix0, mix1, mix0._A_now(xz0), mix1._A_now(xz1))
    return linear_pair(y0, y1, mix0.out_proj.weight, mix1.out_proj.weight)
