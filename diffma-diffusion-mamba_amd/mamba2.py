"""Mamba-2 (SSD) mixer of DiffMa, 'spiral' scan (reference block/mamba2.py:234-457), selected by --use-mamba2; the baseline
orders 'zigma' / 'vim' / 'vmamba' (block/mamba2.py:459-615) run on the same operator with 1 / 2 / 4 row-index tables.

State-dict keys match the reference class: in_proj.weight (2*Din + 2*G*N + H, d_model) in the order
[z | x | B | C | dt], conv1d.weight (Din + 2*G*N, 1, d_conv), conv1d.bias, dt_bias (H), A_log (H), D (H),
norm.weight (Din), out_proj.weight (d_model, Din).

With chunk_size 256 >= L the SSD operator is a single-chunk state-space recurrence whose decay
exp(dt_h * A_h) is a scalar per head (shared by the head's 64 channels and all N states), i.e. exactly the
Mamba-1 recurrence with A[d, n] = A_head(d) and delta[d] = dt_head(d).  The mixer therefore runs on the same
gfx950 kernels (token gather + conv over the 1056 xBC channels, selective scan with the z gather / merge
scatter folded in, token merge); the gated RMSNorm is row-wise and commutes with the token permutation, so the
3-way merge happens right after it and ONE out_proj GEMM follows (out_proj is linear and bias-free).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .selective_scan_interface import linear_splitk, spiral_ssd


class RMSNorm(nn.Module):
    """Gated RMSNorm with the attributes the reference reads (`.weight`, `.eps`; block/mamba2.py:349,402-403)."""

    def __init__(self, hidden_size, eps=1e-5, norm_before_gate=False, group_size=None, device=None, dtype=None):
        super().__init__()
        self.eps, self.norm_before_gate = eps, norm_before_gate
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        if group_size not in (None, hidden_size):
            raise NotImplementedError("grouped RMSNorm (ngroups > 1) is never used by DiffMa")

    def forward(self, x, z=None):
        xf = x.float()
        if z is not None and not self.norm_before_gate:
            xf = xf * F.silu(z.float())
        out = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight.float()
        if z is not None and self.norm_before_gate:
            out = out * F.silu(z.float())
        return out.to(x.dtype)


class Mamba2(nn.Module):
    def __init__(self, d_model, d_state=128, d_conv=4, conv_init=None, expand=2, headdim=64, d_ssm=None, ngroups=1,
                 A_init_range=(1, 16), D_has_hdim=False, rmsnorm=True, norm_before_gate=False, dt_min=0.001, dt_max=0.1,
                 dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), bias=False, conv_bias=True, chunk_size=256,
                 use_mem_eff_path=True, layer_idx=None, process_group=None, sequence_parallel=True, device=None, dtype=None,
                 token_list=(), token_list_reversal=(), origina_list=(), origina_list_reversal=()):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        if process_group is not None:
            raise NotImplementedError("tensor/sequence parallel Mamba-2 is dead code in the reference (process_group is always None)")
        if ngroups != 1 or D_has_hdim or not rmsnorm or (d_ssm not in (None, expand * d_model)):
            raise NotImplementedError("only the configuration DiffMa instantiates is built (ngroups=1, scalar D, rmsnorm, d_ssm=d_inner)")
        self.d_model, self.d_state, self.d_conv, self.expand, self.headdim = d_model, d_state, d_conv, expand, headdim
        self.d_inner = expand * d_model
        self.d_ssm, self.ngroups = self.d_inner, ngroups
        assert self.d_ssm % headdim == 0
        self.nheads = self.d_ssm // headdim
        self.norm_before_gate, self.dt_limit, self.chunk_size = norm_before_gate, dt_limit, chunk_size
        self.rmsnorm, self.activation, self.layer_idx = rmsnorm, "silu", layer_idx
        d_in_proj = 2 * self.d_inner + 2 * ngroups * d_state + self.nheads
        self.in_proj = nn.Linear(d_model, d_in_proj, bias=bias, **fk)
        conv_dim = self.d_ssm + 2 * ngroups * d_state
        self.conv1d = nn.Conv1d(conv_dim, conv_dim, kernel_size=d_conv, groups=conv_dim, padding=d_conv - 1, bias=conv_bias, **fk)
        if conv_init is not None:
            nn.init.uniform_(self.conv1d.weight, -conv_init, conv_init)
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.dt_bias._no_weight_decay = True
        assert 0 < A_init_range[0] <= A_init_range[1]
        self.A_log = nn.Parameter(torch.log(torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)).to(dtype=dtype))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.nheads, device=device))
        self.D._no_weight_decay = True
        self.norm = RMSNorm(self.d_ssm, eps=1e-5, norm_before_gate=norm_before_gate, group_size=self.d_ssm // ngroups, **fk)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

        self.token_list, self.token_list_reversal = list(token_list), list(token_list_reversal)
        self.origina_list, self.origina_list_reversal = list(origina_list), list(origina_list_reversal)
        # token_list is one permutation (spiral, zigma) or a list of four (vmamba, reference model.py:182-186)
        nested = bool(self.token_list) and isinstance(self.token_list[0], (list, tuple))
        L = 0 if nested or not self.token_list_reversal else len(self.token_list)
        idx = torch.tensor([list(range(L)), self.token_list, self.token_list_reversal], dtype=torch.int32) if L else torch.zeros((3, 0), dtype=torch.int32)
        self.register_buffer("scan_index", idx, persistent=False)
        self.register_buffer("scan_index_inv", torch.argsort(idx.long(), dim=1).to(torch.int32), persistent=False)
        self._tables = {}                  # (scan_type, L, device) -> (gather table, inverse) of the baseline scan orders

    def _baseline_tables(self, scan_type, L, device):
        """ZigMa: the block's one permutation; ViM: identity + time reversal (flipped back along the token axis here, unlike
        the Mamba-1 twin); VMamba: four permutations (reference block/mamba2.py:459-615).  'eff' raises in the reference
         and here."""
        key = (scan_type, L, str(device))
        tab = self._tables.get(key)
        if tab is None:
            if scan_type == "zigma":
                rows = [self.token_list]
            elif scan_type == "vim":
                rows = [list(range(L)), list(range(L - 1, -1, -1))]
            elif scan_type == "vmamba":
                rows = self.token_list
            else:
                raise NotImplementedError(f"scan_type={scan_type!r} (the reference's Mamba-2 EfficientVMamba branch raises TypeError)")
            idx = torch.tensor(rows, dtype=torch.int32, device=device)
            if idx.shape[1] != L:
                raise ValueError(f"sequence length {L} != scan table length {idx.shape[1]}")
            tab = (idx, torch.argsort(idx.long(), dim=1).to(torch.int32))
            self._tables[key] = tab
        return tab

    def forward(self, u, scan_type="spiral", seqlen=None, seq_idx=None, inference_params=None):
        """u: (B, L, d_model) -> (B, L, d_model)."""
        if seqlen is not None or seq_idx is not None or inference_params is not None:
            raise NotImplementedError("packed sequences / decode are never used by DiffMa")
        if self.dt_limit != (0.0, float("inf")):
            raise NotImplementedError("dt_limit")
        Bsz, L, _ = u.shape
        Din, N, H, P = self.d_inner, self.d_state, self.nheads, self.headdim
        index, index_inv = (self.scan_index, self.scan_index_inv) if scan_type == "spiral" else self._baseline_tables(scan_type, L, u.device)
        zxbcdt = linear_splitk(u, self.in_proj.weight, self.in_proj.bias)             # [B, L, 2*Din + 2N + H], token-major
        A = -torch.exp(self.A_log.float())                                        # [H]
        y = spiral_ssd(zxbcdt, self.conv1d.weight, self.conv1d.bias, self.dt_bias, A, self.D, self.norm.weight, self.norm.eps,
                       index, index_inv, Din, N)                                     # [B, L, Din]: gated, normalised, merged
        if scan_type == "vim":                                                    # (out1 + out2) / 2, block/mamba2.py:523
            if self.out_proj.bias is not None:
                raise NotImplementedError("out_proj bias with averaged directions")
            y = y * 0.5
        return linear_splitk(y.to(zxbcdt.dtype), self.out_proj.weight, self.out_proj.bias)
