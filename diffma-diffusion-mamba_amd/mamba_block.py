"""Spiral_MambaBlock: the DiffMa block (reference block/mamba_block.py:13-130), and the four baseline blocks the reference
reproduces from other papers on the same mixer (ZigMa, ViM, VMamba, EfficientVMamba: block/mamba_block.py:132-398).

    shift, scale, gate = adaLN(c)                    x_ssm = LN(x)*(1+scale)+shift
    w_ssm = x_ssm * w   (soft mask, w in (0,1))      x_ssm = mamba1(x_ssm); w_ssm = mamba2(w_ssm)
    a = sigmoid(MLP(LN(cat[x_ssm, w_ssm])))          x = x + gate * (a*x_ssm + (1-a)*w_ssm)
Sub-module names equal the reference's, so state dicts are interchangeable.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import block_ops, hip_ops
from .mamba import Mamba, forward_pair
from .selective_scan_interface import GemmChain, _mm_f32, linear_splitk


_SILU_ONCE = [None]


def _silu_once(c, dtype):
    """SiLU(c) in the activation dtype, computed ONCE per conditioning tensor: every block of the denoiser applies the same
    SiLU to the same `c` (reference block/mamba_block.py:82-85,101: adaLN_modulation = Sequential(SiLU, Linear)), which eager
    execution repeats depth times forward and backward (SiLU, cast, cast, SiLU') -- at one sample per GPU those are launches the
    step is made of.  The blocks' gradient contributions are added by autograd.  Keyed on the tensor object and its version."""
    import weakref

    ent = _SILU_ONCE[0]
    grad = torch.is_grad_enabled() and c.requires_grad
    if ent is not None and ent[0]() is c and ent[1] == c._version and ent[2] == dtype and ent[3] == grad:
        out = ent[4]() if grad else ent[4]
        if out is not None:
            return out
    out = torch.nn.functional.silu(c)
    if out.dtype != dtype:
        out = out.to(dtype)
    # With a graph attached the cache holds the result WEAKLY: the blocks' autograd nodes keep it alive until the backward has run,
    # after which a later call with the same tensor recomputes instead of handing out a result whose graph has been freed.
    _SILU_ONCE[0] = (weakref.ref(c), c._version, dtype, grad, weakref.ref(out) if grad else out)
    return out


class _AdaLNAllFn(torch.autograd.Function):
    """The adaLN Linear of EVERY block in one product.  All blocks modulate with the same SiLU(c) (reference block/mamba_block.py:
    82-85, 101; model.py:286-295 passes one `c` down the stack), so their 16 products [B, 2 D] x [2 D, 3 D] are one product with the
    stacked weight [16 * 3 D, 2 D] -- which exists without a copy: step_prep keeps the blocks' 16-bit weight copies as the rows of one
    buffer.  Forward: one GEMM; backward: the blocks' gradients concatenated, one product for d SiLU(c), one for all weight gradients
    (handed back as row views), one column sum for the biases: 5 launches where the per-block form has 4 per block."""

    @staticmethod
    def forward(ctx, sc, wbase, bbase, *masters):
        nb, N, K = wbase.shape
        out = torch.addmm(bbase.view(-1), sc, wbase.view(nb * N, K).t())          # [B, nb * N]
        ctx.save_for_backward(sc, wbase)
        ctx.nb, ctx.N = nb, N
        ctx.dtypes = [m.dtype for m in masters]
        return tuple(out[:, i * N:(i + 1) * N] for i in range(nb))

    @staticmethod
    def backward(ctx, *grads):
        sc, wbase = ctx.saved_tensors
        nb, N = ctx.nb, ctx.N
        K = wbase.shape[-1]
        gs = [g if g is not None else sc.new_zeros((sc.shape[0], N)) for g in grads]
        g_all = torch.cat([g.to(sc.dtype) for g in gs], dim=1)                   # [B, nb * N]
        with torch.autocast(device_type=sc.device.type, enabled=False):
            d_sc = g_all @ wbase.view(nb * N, K) if ctx.needs_input_grad[0] else None
            dW = _mm_f32(g_all.t(), sc).view(nb, N, K)                             # fp32 [nb, N, K]
            db = g_all.sum(0, dtype=torch.float32).view(nb, N)
        dws = [dW[i].to(ctx.dtypes[i]) for i in range(nb)]
        dbs = [db[i].to(ctx.dtypes[nb + i]) for i in range(nb)]
        return (d_sc, None, None, *dws, *dbs)


def adaln_all(model, c):
    """Per-block (shift, scale, gate) triples for the whole stack from one product, or None when the stacked 16-bit weights are not
    current (no prepare() this step: inference, CPU, a weight written since) -- the blocks then run their own adaLN as before."""
    from . import step_prep

    if not (c.is_cuda and torch.is_grad_enabled() and ADALN_ALL) or getattr(model, "_no_adaln_all", False):
        return None
    act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else c.dtype
    if act not in (torch.bfloat16, torch.float16):
        return None
    st, pr = step_prep.adaln_stack(model, act), step_prep.adaln_params(model)
    if st is None or pr is None or len(pr[0]) != len(model.blocks):
        return None
    outs = _AdaLNAllFn.apply(_silu_once(c, act), st[0], st[1], *pr[0], *pr[1])
    return [o.chunk(3, dim=1) for o in outs]


ADALN_ALL = os.environ.get("DIFFMA_ADALN_ALL", "1") == "1"        # 0: one adaLN product per block again (A/B runs)


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class Spiral_MambaBlock(nn.Module):
    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, token_list, token_list_reversal,
                 origina_list, origina_list_reversal, use_mamba2=False):
        super().__init__()
        self.D_dim, self.E_dim, self.dt_rank, self.dim_inner, self.d_state = D_dim, E_dim, dt_rank, dim_inner, d_state
        lists = dict(token_list=token_list, token_list_reversal=token_list_reversal, origina_list=origina_list,
                     origina_list_reversal=origina_list_reversal)
        self.norm1 = nn.LayerNorm(D_dim)
        if use_mamba2:
            from .mamba2 import Mamba2 as mixer
        else:
            mixer = Mamba
        # NB the reference does not forward dt_rank to the mixer: dt_rank = ceil(D_dim/16)
        self.mamba1 = mixer(d_model=D_dim, d_state=d_state, d_conv=4, expand=2, **lists)
        self.mamba2 = mixer(d_model=D_dim, d_state=d_state, d_conv=4, expand=2, **lists)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(2 * D_dim, 3 * D_dim, bias=True))
        self.attention_network = nn.Sequential(nn.LayerNorm(2 * D_dim), nn.Linear(2 * D_dim, D_dim, bias=True), nn.SiLU(),
                                               nn.Linear(D_dim, 1, bias=True), nn.Sigmoid())
        self.initialize_weights()

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for i in (1, 3):
            nn.init.constant_(self.attention_network[i].weight, 0)
            nn.init.constant_(self.attention_network[i].bias, 0)

    fused_elementwise = True      # single-pass HIP kernels for the LN/modulate/mask, LN(cat) and blend chains (GPU only)
    # OPT-IN (DIFFMA_OVERLAP_MIXERS=1): run the block's two mixers on two HIP streams.  They are independent, and each
    # alternates between VALU-bound scans and HBM-bound GEMM / conv / merge kernels, so the two queues fill each other's idle
    # unit (measured: 280 -> 265 ms per DiffMa-L/2 step at batch 512, 22.7 -> 18.5 ms for the graphed batch-8 step); autograd
    # replays every backward node on its forward stream, so the backward overlaps the same way.  Forcing a schedule (scans
    # chained by events, second mixer started at the first one's scan, a high-priority stream) measured 2-3 % slower than
    # letting the queues run free.  NOT the default: the GEMM libraries' persistent stream-K kernels spin-wait for their own
    # not-yet-resident workgroups, and two of them co-scheduled from two queues can starve each other -- DiffMa-XL/2 with
    # Mamba-2 mixers hung the GPU that way, 3 runs of 3 (the DiffMa-L/2 shapes of the recorded solution table never did in ~20
    # runs).  With every GEMM of the two streams chained behind the previous one (GemmChain) the hang is gone, and so is the
    # gain (281 ms): what overlapped profitably were the small GEMMs with each other.  TENSILE_STREAMK_DATA_PARALLEL=1 (set by
    # the package when this mode is requested) removes the spin-waits instead and keeps the gain; 11 clean runs so far.
    overlap_mixers = os.environ.get("DIFFMA_OVERLAP_MIXERS", "0") == "1"
    _side_streams = {}
    _main_streams = {}

    @classmethod
    def _side_stream(cls, device):
        st = cls._side_streams.get(device)
        if st is None:
            st = cls._side_streams[device] = torch.cuda.Stream(device=device)
        return st

    @classmethod
    def join_streams(cls):
        """Make the current stream wait for everything queued on the two mixer streams (used by the DDP communication hook:
        a gradient bucket may hold gradients that were copied in on either stream, the collective only orders itself
        behind the stream that is current when the bucket fills)."""
        for dev, side in cls._side_streams.items():
            cur = torch.cuda.current_stream(dev)
            for st in (side, cls._main_streams.get(dev)):
                if st is not None and st != cur:
                    cur.wait_stream(st)

    def _mixers(self, x_ssm, w_ssm):
        if not (self.overlap_mixers and x_ssm.is_cuda):
            if isinstance(self.mamba1, Mamba):           # small launches: every stage once for both mixers (mamba.forward_pair)
                return forward_pair(self.mamba1, self.mamba2, x_ssm, w_ssm)
            return self.mamba1(x_ssm, "spiral"), self.mamba2(w_ssm, "spiral")
        main = torch.cuda.current_stream(x_ssm.device)
        side = self._side_stream(x_ssm.device)
        self._main_streams[x_ssm.device] = main
        # No two library GEMMs resident together while persistent stream-K kernels are possible.  With hipBLASLt's stream-K in
        # data-parallel form (TENSILE_STREAMK_DATA_PARALLEL=1, set by the package when the mode is requested through the
        # environment) the guard is off: GEMM next to GEMM is where the gain comes from.
        GemmChain.enabled = os.environ.get("DIFFMA_GEMM_CHAIN", "0" if os.environ.get("TENSILE_STREAMK_DATA_PARALLEL") == "1" else "1") != "0"
        side.wait_stream(main)
        with torch.cuda.stream(side):
            w_in = w_ssm
            w_ssm = self.mamba2(w_in, "spiral")
        w_in.record_stream(side)                  # allocated on `main`, read on `side`
        x_ssm = self.mamba1(x_ssm, "spiral")
        main.wait_stream(side)
        w_ssm.record_stream(main)                 # allocated on `side`, read on `main`
        return x_ssm, w_ssm

    def forward(self, x, c, w, mod=None):
        if mod is not None:                       # (shift, scale, gate) of this block from the stack-wide product (adaln_all)
            shift, scale, gate = mod
        elif self.fused_elementwise and x.is_cuda:
            # adaLN through linear_splitk: its 16-bit weight / bias copies come from the step's foreach cast (step_prep) instead of two
            # cast launches per block, and the weight gradient leaves its GEMM in fp32
            ada = self.adaLN_modulation
            act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else c.dtype
            shift, scale, gate = linear_splitk(_silu_once(c, act), ada[1].weight, ada[1].bias).chunk(3, dim=1)
        else:
            shift, scale, gate = self.adaLN_modulation(c).chunk(3, dim=1)
        if self.fused_elementwise and x.is_cuda:
            act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
            block_ops.set_output_dtype(act)
            pt = block_ops.PASSTHROUGH and torch.is_grad_enabled()
            if pt:      # x, x_ssm, w_ssm each feed a LayerNorm node AND the blend: the aliases keep them single-consumer (block_ops)
                x_ssm, w_ssm, x = block_ops.ln_modulate_mask(x, self.norm1, shift, scale, w, passthrough=True)
            else:
                x_ssm, w_ssm = block_ops.ln_modulate_mask(x, self.norm1, shift, scale, w)
            x_ssm, w_ssm = self._mixers(x_ssm, w_ssm)
            net = self.attention_network
            # the two Linear layers of the fusion MLP through linear_splitk: their weight gradients reduce over B*L rows
            if pt:
                hcat, x_ssm, w_ssm = block_ops.ln_cat(x_ssm, w_ssm, net[0], passthrough=True)
            else:
                hcat = block_ops.ln_cat(x_ssm, w_ssm, net[0])
            if block_ops.GATE_HEAD and hip_ops.gate_head_supported(hcat, net[1].out_features):
                # bias + SiLU + Linear(C, 1) + Sigmoid ride on ONE read of the first Linear's output (csrc/gate_head.hip)
                a = block_ops.gate_head(linear_splitk(hcat, net[1].weight, None), net[1].bias, net[3].weight, net[3].bias)
            else:
                a = net[4](linear_splitk(net[2](linear_splitk(hcat, net[1].weight, net[1].bias)), net[3].weight, net[3].bias))
            return block_ops.blend_residual(x, x_ssm, w_ssm, a, gate)
        x_ssm = modulate(self.norm1(x), shift, scale)
        w_ssm = x_ssm * w
        x_ssm = self.mamba1(x_ssm, "spiral")
        w_ssm = self.mamba2(w_ssm, "spiral")
        a = self.attention_network(torch.cat([x_ssm, w_ssm], dim=-1))
        return x + gate.unsqueeze(1) * (a * x_ssm + (1 - a) * w_ssm)


class _BaselineMambaBlock(nn.Module):
    """adaLN -> LN/modulate -> ONE mixer with the block's scan order -> gated residual (reference block/mamba_block.py:
    190-198, 247-255, 318-326, 381-389: the four forward() bodies differ only in the scan type).  `w` (the soft mask) is
    accepted and ignored, as in the reference.  Sub-module names equal the reference's (norm1, adaLN_modulation, mamba)."""
    scan_type = None
    mixer_first = False

    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2=False, token_list=(), origina_list=()):
        super().__init__()
        if use_mamba2 and self.scan_type == "eff":
            raise NotImplementedError("EfficientVMamba on the Mamba-2 mixer raises TypeError in the reference too (SURVEY.md A.4-7)")
        self.D_dim, self.E_dim, self.dt_rank, self.dim_inner, self.d_state = D_dim, E_dim, dt_rank, dim_inner, d_state
        self.token_list, self.origina_list = token_list, origina_list
        if use_mamba2:
            from .mamba2 import Mamba2 as mixer_cls
        else:
            mixer_cls = Mamba
        mixer = lambda: mixer_cls(d_model=D_dim, d_state=d_state, d_conv=4, expand=2, token_list=token_list, origina_list=origina_list)
        if self.mixer_first:                   # registration order = state-dict key order of the reference class
            self.mamba = mixer()
        self.norm1 = nn.LayerNorm(D_dim)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(2 * D_dim, 3 * D_dim, bias=True))
        if not self.mixer_first:
            self.mamba = mixer()
        self.initialize_weights()

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x, c, w=None):
        shift, scale, gate = self.adaLN_modulation(c).chunk(3, dim=1)
        x_ssm = self.mamba(modulate(self.norm1(x), shift, scale), self.scan_type)
        return x + gate.unsqueeze(1) * x_ssm


class Zig_MambaBlock(_BaselineMambaBlock):
    scan_type = "zigma"

    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, token_list, origina_list, use_mamba2=False):
        super().__init__(D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2, token_list, origina_list)


class ViM_MambaBlock(_BaselineMambaBlock):
    scan_type = "vim"
    mixer_first = True

    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2=False):
        super().__init__(D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2)


class VMamba_MambaBlock(_BaselineMambaBlock):
    scan_type = "vmamba"
    mixer_first = True

    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, token_list, origina_list, use_mamba2=False):
        super().__init__(D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2, token_list, origina_list)


class EfficientVMamba_MambaBlock(_BaselineMambaBlock):
    scan_type = "eff"

    def __init__(self, D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2=False):
        super().__init__(D_dim, E_dim, dt_rank, dim_inner, d_state, use_mamba2)


def make_ddp_comm_hook(grad_compression="none", join_streams=False):
    """DDP communication hook factory (train.wrap_ddp): optionally join the two mixer streams first, then all-reduce the bucket
    (mean over ranks) either as it is (fp32, the reference's behaviour) or as a bf16 / fp16 copy that is cast back afterwards."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

    reduce_hook = {"none": default_hooks.allreduce_hook, "bf16": default_hooks.bf16_compress_hook,
                   "fp16": default_hooks.fp16_compress_hook}[grad_compression]

    def hook(process_group, bucket):
        if join_streams:
            Spiral_MambaBlock.join_streams()
        return reduce_hook(process_group, bucket)

    return hook


def ddp_join_streams_hook(process_group, bucket):
    """DDP communication hook for models whose blocks run their mixers on two streams: join the streams, then the stock
    all-reduce (mean over ranks).  Register with `ddp.register_comm_hook(process_group_or_None, ddp_join_streams_hook)`."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

    Spiral_MambaBlock.join_streams()
    return default_hooks.allreduce_hook(process_group, bucket)
