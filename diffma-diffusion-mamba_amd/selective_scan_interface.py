"""Reference-facing operators, backed by the gfx950 kernels (C ABI: include/diffma_hip.h).

Names, argument order and meaning mirror what the reference imports
    block/mamba.py:11   from mamba_ssm.ops.selective_scan_interface import selective_scan_fn, mamba_inner_fn
    block/mamba.py:13   from causal_conv1d import causal_conv1d_fn
so a maintainer can point those imports here.  Tensors arrive in the reference's
channel-major convention (B, D, L); internally everything is token-major [B, L, D] (channel stride 1),
which is what the lane-per-channel kernels want.  A (B, D, L)-shaped *view* of a token-major buffer is
consumed without a copy; a genuinely L-contiguous tensor is repacked once.

`spiral_ssm` is the fused 3-direction operator the DiffMa mixer uses (block/mamba.py:343-355): token
gather + conv1d + SiLU, x_proj, dt_proj, selective scan with the z gather and the CrossMerge
inverse reindex folded into row addressing, then the 3-way merge BEFORE out_proj.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import hip_ops
from .step_prep import cast_weight, stacked_pair


REPACK_OWN = os.environ.get("DIFFMA_REPACK_OWN", "1") == "1"        # 0: the ATen strided copy again (A/B runs)


def _repack_own(t: torch.Tensor) -> bool:
    """dm_repack takes the tensor: a contiguous-last-axis 3-d device tensor of a supported dtype whose batch fits the launch grid's
    z extent (65 535; larger stacked batches keep the ATen strided copy, which has no such limit)."""
    return (REPACK_OWN and t.is_cuda and t.dim() == 3 and t.stride(2) == 1 and t.shape[0] <= 65535
            and t.dtype in (torch.float32, torch.bfloat16, torch.float16))


def _to_token_major(t: torch.Tensor) -> torch.Tensor:
    """(B, D, L) in any layout -> a [B, L, D] tensor with stride(-1) == 1 (no copy when possible; a genuinely L-contiguous
    tensor -- what the reference hands over, block/mamba.py:333-348 -- goes through dm_repack).  No autograd: callers are
    autograd Functions' forward / backward bodies; `_RepackFn` is the differentiable form."""
    tm = t.transpose(1, 2)
    if tm.stride(-1) == 1:
        return tm
    if _repack_own(t):
        return hip_ops.repack(t, True)
    return tm.contiguous()


def _to_channel_major(t: torch.Tensor) -> torch.Tensor:
    """[B, L, D] token-major -> a CONTIGUOUS (B, D, L) tensor (the layout the reference's glue goes on with: CrossScan.backward,
    the in_proj gradient products, block/mamba.py:47-57, 333-337); a lazily transposed view would make every consumer a
    strided ATen copy."""
    if _repack_own(t):
        return hip_ops.repack(t, False)
    return t.transpose(1, 2).contiguous()


class _RepackFn(torch.autograd.Function):
    """(B, D, L) channel-major -> [B, L, D] token-major with the gradient handed back channel-major and contiguous."""

    @staticmethod
    def forward(ctx, t):
        return _to_token_major(t)

    @staticmethod
    def backward(ctx, g):
        if g.stride(-1) != 1:
            g = g.contiguous()
        return _to_channel_major(g)


def _bc_token_major(Bm: torch.Tensor):
    """B/C arrive as (B, N, L) or (B, G, N, L); return ([B, L, G*N] with state stride 1, G)."""
    if Bm.dim() == 3:
        Bm = Bm[:, None]
    bsz, G, N, L = Bm.shape
    t = Bm.permute(0, 3, 1, 2)                                  # (B, L, G, N)
    if not (t.stride(3) == 1 and t.stride(2) == N):
        t = t.contiguous()
    return t.reshape(bsz, L, G * N) if t.is_contiguous() else t.as_strided((bsz, L, G * N), (t.stride(0), t.stride(1), 1)), G


class _SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, return_last_state, grad_on=True):
        ut, dt = _to_token_major(u), _to_token_major(delta)
        zt = _to_token_major(z) if z is not None else None
        if dt.dtype != ut.dtype:
            dt = dt.to(ut.dtype)
        if zt is not None and zt.dtype != ut.dtype:
            zt = zt.to(ut.dtype)
        Bt, G = _bc_token_major(Bm)
        Ct, _ = _bc_token_major(Cm)
        if Bt.dtype not in (torch.float32, ut.dtype):
            Bt = Bt.to(ut.dtype)
        if Ct.dtype != Bt.dtype:
            Ct = Ct.to(Bt.dtype)
        S, L, Dm = ut.shape
        N = A.shape[1]
        # grad_on = the caller's grad mode (inside forward() it is always off): no checkpoints under torch.no_grad()
        need_grad = grad_on and any(t is not None and t.requires_grad for t in (u, delta, A, Bm, Cm, D, z, delta_bias))
        ckpt = None
        if need_grad:
            ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, ut.dtype, ut.device)
        last = torch.empty((S, N, Dm), dtype=torch.float32, device=ut.device) if return_last_state else None
        out = hip_ops.scan_fwd(ut, dt, A, Bt, Ct, D, zt, delta_bias, delta_softplus, ckpt=ckpt, last_state=last,
                               ngroups=G)
        ctx.delta_softplus = delta_softplus
        ctx.G = G
        ctx.has_z, ctx.has_D, ctx.has_bias = z is not None, D is not None, delta_bias is not None
        ctx.b4, ctx.c4 = Bm.dim() == 4, Cm.dim() == 4
        ctx.save_for_backward(ut, dt, A, Bt, Ct, D, zt, delta_bias, ckpt)
        out_cm = out.transpose(1, 2)                            # (B, D, L) view, like the reference returns
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out_cm, last.transpose(1, 2)
        return out_cm

    @staticmethod
    def backward(ctx, dout, *ignored):
        ut, dt, A, Bt, Ct, D, zt, delta_bias, ckpt = ctx.saved_tensors
        if ctx.G != 1:
            raise NotImplementedError("backward with ngroups > 1 is not wired (DiffMa uses ngroups = 1)")
        dot = _to_token_major(dout)
        if dot.dtype != ut.dtype:
            dot = dot.to(ut.dtype)
        du, ddelta, dz, dB, dC, dA, dD, dbias = hip_ops.scan_bwd(ut, dt, A, Bt, Ct, D, zt, delta_bias, dot, ckpt,
                                                                 ctx.delta_softplus)
        cm = lambda t: t.transpose(1, 2)
        dBm = dB.to(Bt.dtype).transpose(1, 2)                   # (B, N, L)
        dCm = dC.to(Ct.dtype).transpose(1, 2)
        if ctx.b4:
            dBm = dBm[:, None]
        if ctx.c4:
            dCm = dCm[:, None]
        return (cm(du), cm(ddelta), dA.to(A.dtype), dBm, dCm,
                dD.to(D.dtype) if ctx.has_D else None, cm(dz) if ctx.has_z else None,
                dbias.to(delta_bias.dtype) if ctx.has_bias else None, None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """Drop-in for mamba_ssm's selective_scan_fn (imported at block/mamba.py:11).

    u, delta, z: (B, D, L); A: (D, N) real; B, C: (B, N, L) or (B, G, N, L) (input-dependent);
    D, delta_bias: (D,) fp32.  Returns out (B, D, L) [, last_state (B, D, N)].
    """
    if A.is_complex():
        raise NotImplementedError("complex A is not supported (DiffMa uses the real S4D-real init, block/mamba.py:304-310)")
    if B.dim() not in (3, 4) or C.dim() not in (3, 4):
        raise NotImplementedError("only input-dependent B/C are supported (the only mode DiffMa uses)")
    return _SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state, torch.is_grad_enabled())


class _CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, silu):
        xt = _to_token_major(x)
        out = hip_ops.gather_conv1d_fwd(xt, weight, bias, silu=silu)
        ctx.silu = silu
        ctx.save_for_backward(xt, weight, bias)
        return out.transpose(1, 2)

    @staticmethod
    def backward(ctx, dout):
        xt, weight, bias = ctx.saved_tensors
        dot = _to_token_major(dout)
        if dot.dtype != xt.dtype:
            dot = dot.to(xt.dtype)
        dx, dw, db = hip_ops.gather_conv1d_bwd(xt, weight, bias, dot, silu=ctx.silu)
        return dx.transpose(1, 2), dw.to(weight.dtype).reshape(weight.shape), (db.to(bias.dtype) if bias is not None else None), None


def causal_conv1d_fn(x, weight, bias=None, seq_idx=None, initial_states=None, return_final_states=False,
                     final_states_out=None, activation=None):
    """Drop-in for causal_conv1d.causal_conv1d_fn (imported at block/mamba.py:13).  x: (B, D, L), weight (D, W)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu or swish")
    if seq_idx is not None or initial_states is not None or return_final_states or final_states_out is not None:
        raise NotImplementedError("seq_idx / initial_states / final_states are decode-time features DiffMa never uses")
    return _CausalConv1dFn.apply(x, weight, bias, activation is not None)


class GemmChain:
    """Used only by the opt-in two-stream mode of the block (mamba_block.Spiral_MambaBlock._mixers): every GEMM of the two
    mixer streams is chained behind the previous one with an event, so that no two library GEMMs are ever resident together.
    hipBLASLt / Tensile's persistent stream-K kernels spin-wait for their own not-yet-scheduled workgroups; two of them
    co-scheduled from two queues can starve each other (observed as a GPU hang).  GEMM next to scan / conv / merge kernels
    stays concurrent -- those workgroups always retire."""
    enabled = False
    event = None
    stream = None

    @classmethod
    def run(cls, fn, *args, **kw):
        if not cls.enabled or not torch.cuda.is_available():
            return fn(*args, **kw)
        cur = torch.cuda.current_stream()
        if cls.event is not None and cls.stream != cur:
            cur.wait_event(cls.event)
        out = fn(*args, **kw)
        cls.event = torch.cuda.Event()
        cls.event.record(cur)
        cls.stream = cur
        return out


def _tn_splitk(a, b):
    return GemmChain.run(_tn_splitk_impl, a, b)


_MM_F32 = [None]          # does torch.mm take out_dtype on this device?  (aten::mm.dtype: 16-bit operands, fp32 result, one launch)


def _mm_f32(x, y):
    """x @ y in fp32 from 16-bit operands WITHOUT a separate cast launch where the library offers it (hipBLASLt accumulates in fp32
    anyway; the 16-bit result + `.float()` of the plain form rounds the weight gradient once more and costs a launch per GEMM -- ~150 per
    training step, which is what a small-batch step is made of)."""
    if x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and _MM_F32[0] is not False and os.environ.get("DIFFMA_MM_F32", "1") == "1":
        try:
            out = torch.mm(x, y, out_dtype=torch.float32)
            _MM_F32[0] = True
            return out
        except (RuntimeError, NotImplementedError, TypeError):
            if _MM_F32[0]:                       # it worked before: a real error, not a missing feature
                raise
            _MM_F32[0] = False
    return (x @ y).float()


def _tn_splitk_impl(a, b):
    """a^T @ b for tall operands a (M, P), b (M, Q): the reduction runs over M = B*L or ndir*B*L (50 176 /
    150 528 at the bench shape), which hipBLASLt does not split -- a single-pass GEMM is 2-8x off its memory
    time (x_proj: 492 us vs 82 us; in_proj: 194 us vs 128 us even after solution tuning; tools/bench_gemm.py).
    Split K into C slabs with one bmm and add the slabs in fp32; C is bounded so that the slab outputs stay
    small next to the operands."""
    M = a.shape[0]

    def slabs(t, C):
        """[M, P] (row stride >= P: a column block of a wider row-major buffer is fine, bmm takes the leading dimension) -> [C, M/C, P]"""
        return t.as_strided((C, M // C, t.shape[1]), (M // C * t.stride(0), t.stride(0), 1))

    for C in (64, 32, 16, 8):
        # (below ~12k rows one GEMM is as fast as the slab product + its sum, and it is one launch instead of two: the small-batch step
        #  is made of launches)
        if M >= 12288 and M % C == 0 and M // C >= 256 and C * a.shape[1] * b.shape[1] <= (1 << 23) and a.stride(1) == 1 and b.stride(1) == 1:
            return torch.bmm(slabs(a, C).transpose(1, 2), slabs(b, C)).sum(0, dtype=torch.float32)      # the cast is fused into the reduction
    return _mm_f32(a.t(), b)


def _own_single(a, b, a_kmajor, b_kmajor, out_dtype=None):
    """dm_gemm for ONE product: same row bound as the paired form, and only for matrices with at least 64 rows (below that a GEMM
    is a handful of workgroups either way)."""
    rows = a.shape[0]
    return (PAIR_GEMM == "own" and 64 <= rows <= PAIR_OWN_MAX_ROWS and a.dim() == 2 and b.dim() == 2
            and hip_ops.gemm_supported(a, b, a_kmajor, b_kmajor, out_dtype))


# Large-batch projections on K12 (csrc/gemm_large.hip).  DIFFMA_GEMM_LARGE=0 (hip_ops.GEMM_LARGE) returns every product to the library.
# DIFFMA_GEMM_LARGE_MIN_COLS: K12 is taken from this many output columns.  Measured against the TunableOp-tuned library at M = 100 352
# (profiles/r05_gemm_large_ablation.txt, stand-alone): N = 2048 and 1024 level or ahead (in_proj forward 216 vs 222 us, out_proj
# input gradient 108 vs 125 / 148 us), N = 512 level (out_proj forward 104 vs 105) or behind (in_proj input gradient, K = 2048: 223 vs
# 198).  Inside the training step K12's launches run ~10 % longer than stand-alone and the step is 0.5 % SLOWER with K12 on the two
# wide products than with the library everywhere (220.1 vs 219.1 ms, same box; 221.9 with K12 on all four): at parity, not ahead.
# The default keeps it on the wide products; DIFFMA_GEMM_LARGE_MIN_COLS=512 puts all four on it.
LARGE_MIN_ROWS = int(os.environ.get("DIFFMA_GEMM_LARGE_MIN_ROWS", "12288"))
LARGE_MIN_COLS = int(os.environ.get("DIFFMA_GEMM_LARGE_MIN_COLS", "1024"))
LARGE_MAX_K_NARROW = int(os.environ.get("DIFFMA_GEMM_LARGE_MAX_K_NARROW", "1024"))     # for outputs narrower than 1024 columns
# dx = dy W as dy (W^T)^T with a transposed 16-bit weight copy: the library's NT kernels beat its NN kernels at these shapes
LARGE_DGRAD_NT = os.environ.get("DIFFMA_DGRAD_NT", "1") == "1"


def _own_large(a, b):
    """dm_gemm_large for C = a @ b^T (a [M, K], b [N, K], both 16-bit, K contiguous)?"""
    M, K = a.shape
    N = b.shape[0]
    if M < LARGE_MIN_ROWS or N < LARGE_MIN_COLS or (N < 1024 and K > LARGE_MAX_K_NARROW):
        return False
    return hip_ops.gemm_large_supported(a, b)


class _LinearSplitKFn(torch.autograd.Function):
    """F.linear whose weight gradient uses the split-K product above (the projections' dW GEMMs have K = B*L)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        dev = x.device.type
        dt_ = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x.dtype
        xc = x if x.dtype == dt_ else x.to(dt_)
        wc = cast_weight(weight, dt_)                    # the step's shadow copy when current (step_prep), else a cast
        ctx.save_for_backward(xc, wc)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        if bias is None and xc.is_cuda and _own_single(xc.reshape(-1, xc.shape[-1]), wc, True, True):
            # small launches: the own kernel is faster than the library's quantised tile grid (csrc/gemm.hip)
            return hip_ops.gemm(xc.reshape(-1, xc.shape[-1]), wc).view(*xc.shape[:-1], wc.shape[0])
        if bias is None and xc.is_cuda and _own_large(xc.reshape(-1, xc.shape[-1]), wc):
            # large batch: the persistent 256 x 256 kernel (csrc/gemm_large.hip, K12)
            return hip_ops.gemm_large(xc.reshape(-1, xc.shape[-1]), wc).view(*xc.shape[:-1], wc.shape[0])
        return GemmChain.run(F.linear, xc, wc, None if bias is None else cast_weight(bias, dt_))

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        x_dt, w_dt, b_dt = ctx.meta
        with torch.autocast(device_type=xc.device.type, enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1])
            if dy2.dtype != xc.dtype:
                dy2 = dy2.to(xc.dtype)
            x2 = xc.reshape(-1, xc.shape[-1])
            dy2c, x2c = dy2.contiguous(), x2.contiguous()
            if not ctx.needs_input_grad[0]:
                dx = None
            elif b_dt is None and _own_single(dy2c, wc, True, False):
                dx = hip_ops.gemm(dy2c, wc, True, False).view(xc.shape).to(x_dt)
            elif LARGE_DGRAD_NT and dy2c.is_cuda and dy2c.shape[0] >= LARGE_MIN_ROWS and dy2c.dtype in (torch.bfloat16, torch.float16):
                # dx = dy W as an "NT" product with a transposed 16-bit copy of the weight (a few MB, one small launch): both
                # operands then have the contraction index contiguous.  Measured at M = 100 352 (tools/bench_gemm_large.py): the
                # library's NN form 217 / 148 us (in_proj / out_proj) against 198 / 125 us for its NT form and 223 / 108 us for K12.
                wt = wc.t().contiguous()
                if _own_large(dy2c, wt):
                    dx = hip_ops.gemm_large(dy2c, wt).view(xc.shape).to(x_dt)
                else:
                    dx = GemmChain.run(F.linear, dy2c, wt).view(xc.shape).to(x_dt)
            else:
                dx = GemmChain.run(torch.mm, dy2, wc).view(xc.shape).to(x_dt)
            if not ctx.needs_input_grad[1]:
                dw = None
            elif b_dt is None and _own_single(dy2c, x2c, False, False, torch.float32):
                dw = hip_ops.gemm(dy2c, x2c, False, False, out_dtype=torch.float32).to(w_dt)
            else:
                dw = _tn_splitk(dy2c, x2c).to(w_dt)
            db = dy2.sum(0, dtype=torch.float32).to(b_dt) if (b_dt is not None and ctx.needs_input_grad[2]) else None
        return dx, dw, db


def linear_splitk(x, weight, bias=None):
    """Drop-in for F.linear(x, weight, bias) on the token-major projections (in_proj / out_proj)."""
    return _LinearSplitKFn.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------
# Fused 3-direction operator of the DiffMa mixer
# ------------------------------------------------------------------------------------------------
# DIFFMA_HOIST_GATE=0: gate (and softplus) back inside every per-direction scan, as upstream evaluates them (A/B runs, tests)
HOIST_GATE = os.environ.get("DIFFMA_HOIST_GATE", "1") == "1"


class _SpiralSSMFn(torch.autograd.Function):
    """xz [B, L, 2*Din] token-major -> merged pre-projection output y [B, L, Din].

    scan_index [ndir, L]: position l of direction k reads token scan_index[k][l]   (CrossScan, block/mamba.py:41-45)
    Because CrossMerge adds direction k's output row origina_k[t] at token t and origina_k is the inverse
    permutation of scan_index[k] (tools.py:38-42), step l's result belongs to token scan_index[k][l]: the
    same table drives the gather of x and z and the scatter of y.
    """

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, Wx, Wdt, dt_bias, A, Dskip, scan_index, grad_on=True, out_index=None, merge=True):
        """out_index (default: scan_index): row of the per-direction output that step l's result is written to -- the
        baseline blocks need a scatter that is not the gather's inverse (ViM, block/mamba.py:362-367).  merge=False returns
        the per-direction outputs [ndir, B, L, Din] instead of their sum."""
        Bsz, L, D2 = xz.shape
        Din = D2 // 2
        ndir = scan_index.shape[0]
        R = Wdt.shape[1]
        N = A.shape[1]
        dt_ = xz.dtype
        x_view, z_view = xz[..., :Din], xz[..., Din:]
        need_grad = grad_on and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[1:8]))   # grad_on: the caller's grad mode
        Wx_c, Wdt_c = cast_weight(Wx, dt_), cast_weight(Wdt, dt_)              # kept for the backward (the step's shadows, or one cast per step)
        if hip_ops.conv_xproj_supported(x_view, Wx_c, ndir * Bsz, conv_w.shape[-1]):
            # gather + conv + SiLU + x_proj in one kernel: x~ is projected while its tile is still on the CU
            xc, x_dbl = hip_ops.gather_conv1d_xproj_fwd(x_view, conv_w, conv_b, Wx_c, row_index=scan_index, ndir=ndir, silu=True)
        else:
            xc = hip_ops.gather_conv1d_fwd(x_view, conv_w, conv_b, row_index=scan_index, ndir=ndir, silu=True)   # [ndir*B, L, Din]
            x_dbl = GemmChain.run(F.linear, xc.view(-1, Din), Wx_c)            # [ndir*B*L, R+2N]
        # Hoisted gate: CrossScan permutes x and z together and CrossMerge applies the inverse permutation (block/mamba.py:41-45,
        # 66-68), so merged[t] = silu(z[t]) * sum_k y~_k[t]: the scans run WITHOUT z and the gate is applied once per token by the
        # merge (one z read in an HBM-bound kernel) instead of three times inside the VALU-bound scans.  Needs the scatter table
        # to be the gather table (not ViM's) and a merge to ride on.
        # One direction (the reference's own three-calls-per-mixer pattern through mamba_inner_fn, the ZigMa order): the gate pass is an
        # extra launch, worth it only where the scans are VALU-bound -- the large launches (K2 1 058 -> ~830 us per 512 sequences).
        hoist = HOIST_GATE and merge and out_index is None and (ndir > 1 or Bsz >= hip_ops.XPROJ_FUSED_MIN_SEQS)
        # Hoisted softplus: delta = softplus(dt_proj(.) + bias) leaves the dt_proj kernel activated (csrc/dtproj.hip) and the scans
        # run with DM_FLAG_DELTA_ACTIVATED (forward: nothing to evaluate; backward: only 1 - exp(-delta)).
        # The backward of that flag exists for d_state 16 only (csrc/scan_bwd.hip): other widths keep the softplus inside the scans.
        act = hoist and N == 16 and hip_ops.dtproj_softplus_supported(x_dbl, Wdt_c)
        if act:
            delta = hip_ops.dtproj_softplus_fwd(x_dbl, Wdt_c, dt_bias).view(ndir * Bsz, L, Din)
        else:
            delta = GemmChain.run(F.linear, x_dbl[:, :R], Wdt_c).view(ndir * Bsz, L, Din)
        xd3 = x_dbl.view(ndir * Bsz, L, R + 2 * N)
        Bm, Cm = xd3[..., R:R + N], xd3[..., R + N:]
        ckpt = None
        if need_grad:
            ckpt = hip_ops.alloc_scan_ckpt(ndir * Bsz, L, N, Din, xz.dtype, xz.device)
        oidx = scan_index if out_index is None else out_index
        ctx.merge, ctx.hoist, ctx.act = merge, hoist, act
        if hoist:
            ydir = hip_ops.scan_fwd(xc, delta, A, Bm, Cm, Dskip, None, dt_bias, True, z_row_index=scan_index, out_row_index=oidx,
                                    batch_per_dir=Bsz, ckpt=ckpt, delta_activated=act)          # token order, NOT gated
            pre = torch.empty((Bsz, L, Din), dtype=dt_, device=xz.device) if need_grad else None   # ungated sum, kept for dz
            y = hip_ops.token_merge(ydir.view(ndir, Bsz, L, Din), gate=z_view, pre_out=pre)
            ctx.save_for_backward(xz, conv_w, conv_b, Wx, Wdt, dt_bias, A, Dskip, scan_index, xc, x_dbl, delta, ckpt, Wx_c, Wdt_c, oidx, pre)
            return y
        ydir = hip_ops.scan_fwd(xc, delta, A, Bm, Cm, Dskip, z_view, dt_bias, True, z_row_index=scan_index,
                                out_row_index=oidx, batch_per_dir=Bsz, ckpt=ckpt)                   # token order
        ctx.save_for_backward(xz, conv_w, conv_b, Wx, Wdt, dt_bias, A, Dskip, scan_index, xc, x_dbl, delta, ckpt, Wx_c, Wdt_c, oidx, None)
        if not merge:
            return ydir.view(ndir, Bsz, L, Din)
        return hip_ops.token_merge(ydir.view(ndir, Bsz, L, Din)) if ndir > 1 else ydir

    @staticmethod
    def backward(ctx, dy):
        xz, conv_w, conv_b, Wx, Wdt, dt_bias, A, Dskip, scan_index, xc, x_dbl, delta, ckpt, Wx_c, Wdt_c, oidx, pre = ctx.saved_tensors
        Bsz, L, D2 = xz.shape
        Din = D2 // 2
        ndir = scan_index.shape[0]
        R = Wdt.shape[1]
        N = A.shape[1]
        dt_ = xz.dtype
        dy = dy.contiguous()
        if dy.dtype != dt_:
            dy = dy.to(dt_)
        if not ctx.merge:
            dy = dy.view(ndir * Bsz, L, Din)           # one gradient per direction
        xd3 = x_dbl.view(ndir * Bsz, L, R + 2 * N)
        Bm, Cm = xd3[..., R:R + N], xd3[..., R + N:]
        z_view = xz[..., Din:]
        M = ndir * Bsz * L
        dx_dbl = torch.empty((M, R + 2 * N), dtype=dt_, device=xz.device)
        dxz = torch.empty_like(xz)
        if ctx.hoist:
            # the gate's backward once per token: g = dy * silu(z) is what the three directions read as their output gradient,
            # dz = dy * pre * silu'(z) goes straight into the z half of d(xz) (no dz slabs, no 3-slab merge)
            g, dz = hip_ops.gate_bwd(dy, z_view, pre, dz_out=dxz[..., Din:])
            du, ddelta, _, _, _, dA, dD, dbias = hip_ops.scan_bwd(
                xc, delta, A, Bm, Cm, Dskip, None, dt_bias, g, ckpt, True, z_row_index=scan_index, out_row_index=oidx,
                batch_per_dir=Bsz, dbc_out=dx_dbl.view(ndir * Bsz, L, R + 2 * N)[..., R:], delta_activated=ctx.act)
        else:
            du, ddelta, dz, _, _, dA, dD, dbias = hip_ops.scan_bwd(
                xc, delta, A, Bm, Cm, Dskip, z_view, dt_bias, dy, ckpt, True, z_row_index=scan_index,
                out_row_index=oidx, batch_per_dir=Bsz, dout_per_seq=not ctx.merge,
                dbc_out=dx_dbl.view(ndir * Bsz, L, R + 2 * N)[..., R:])      # dB | dC land in their x_dbl columns
        ddelta2 = ddelta.view(M, Din)
        if hip_ops.dtproj_bwd_supported(ddelta2, x_dbl, Wdt_c, dx_dbl):
            # both consumers of ddelta -- the dt columns of d(x_dbl) and dW_dt -- in ONE read of it (csrc/dtproj.hip, K8b)
            dWdt = hip_ops.dtproj_bwd(ddelta2, x_dbl, Wdt_c, dx_dbl).to(Wdt.dtype)
        else:
            dx_dbl[:, :R] = GemmChain.run(torch.mm, ddelta2, Wdt_c)        # (a strided `out=` is an untuned GEMM shape class: pathologically slow by default)
            dWdt = _tn_splitk(ddelta2, x_dbl[:, :R]).to(Wdt.dtype)               # [Din, R]; the dt columns of x_dbl in place (leading dimension R + 2N)
        dWx = _tn_splitk(dx_dbl, xc.view(M, Din)).to(Wx.dtype)                   # [R+2N, Din]
        if hip_ops.conv_xproj_bwd_supported(xz[..., :Din], Wx_c, ndir * Bsz, conv_w.shape[-1], du, dx_dbl):
            # d x~ = du + dx_dbl @ Wx is formed tile by tile inside the conv backward (K4x) instead of by an addmm over [M, Din]
            merged_dx = hip_ops.DX_MERGED and conv_w.shape[-1] == 4         # one direction: dx straight into d(xz), no copy pass
            dx_slabs, dconv_w, dconv_b = hip_ops.gather_conv1d_xproj_bwd(xz[..., :Din], conv_w, conv_b, du, dx_dbl, Wx_c.t().contiguous(),
                                                                         row_index=scan_index, ndir=ndir, silu=True,
                                                                         merged_out=dxz[..., :Din] if merged_dx else None)
        else:
            merged_dx = False
            # in place: an out-of-place addmm first copies `du` into its result (a 2 x 308 MB device memcpy per call)
            dxc = GemmChain.run(du.view(M, Din).addmm_, dx_dbl, Wx_c).view(ndir * Bsz, L, Din)
            dx_slabs, dconv_w, dconv_b = hip_ops.gather_conv1d_bwd(xz[..., :Din], conv_w, conv_b, dxc,
                                                                   row_index=scan_index, ndir=ndir, silu=True)
        if not merged_dx:                                          # K4x already summed the directions into dxz[..., :Din]
            hip_ops.token_merge(dx_slabs.view(ndir, Bsz, L, Din), out=dxz[..., :Din])
        if not ctx.hoist:
            hip_ops.token_merge(dz.view(ndir, Bsz, L, Din), out=dxz[..., Din:])
        return (dxz, dconv_w.to(conv_w.dtype).reshape(conv_w.shape), dconv_b.to(conv_b.dtype) if conv_b is not None else None,
                dWx, dWdt, dbias.to(dt_bias.dtype), dA.to(A.dtype), dD.to(Dskip.dtype), None, None, None, None)


# ------------------------------------------------------------------------------------------------
# The TWO mixers of a DiffMa block as one set of launches (reference block/mamba_block.py:107-108)
# ------------------------------------------------------------------------------------------------
# The reference runs mamba1(x_ssm) and mamba2(w_ssm) one after the other: same shapes, different weights and spiral tables.  Its own
# configuration trains at ONE sample per GPU (config/brain.yaml:11), where a step is bound by the number of kernel launches (eager:
# the host's launch rate; hipGraph: ~8 us of dispatch per node), not by what the kernels do.  The pair path issues every stage once
# for both mixers: the projections as batched GEMMs over stacked weights, the kernels through hip_ops.paired() -> the `_n` entry
# points of the C ABI (one grid, blockIdx.z picks the mixer).  Arithmetic per mixer is unchanged: kernels bit-identical, GEMMs to
# library rounding.  Used for small launches only (below the fused conv + x_proj threshold); large batches gain nothing from it.
PAIR_MIXERS = os.environ.get("DIFFMA_PAIR_MIXERS", "1") == "1"

# How the pair path multiplies by the two mixers' projection weights.
#   "own" (default): dm_gemm (csrc/gemm.hip), ONE launch for both mixers through hip_ops.paired() -- up to PAIR_OWN_MAX_ROWS rows per
#          mixer, where it is also faster on the device than the library's two GEMMs (DiffMa-L/2 widths, both mixers, us incl. launch,
#          tools/bench_gemm_own.py: batch 8 in_proj 43 -> 21 forward, 42 -> 30 weight gradient, out_proj 41 -> 16 / 38 -> 15 / 40 -> 22;
#          from ~6 000 rows the library's large tiles win); above that, and for shapes dm_gemm does not take, "mm".
#   "mm":  one plain library GEMM per mixer -- the products the unpaired path has always issued, tuned and recorded.
#   "bmm": ONE batched library GEMM per product (batch = 2), TunableOp's first-use tuning switched off around the call.  NOT safe:
#          torch.bmm([2, 12544, 1024] x [2, 1024, 512]) -- out_proj at batch 64 -- returns NaN / faults with the library's DEFAULT kernel
#          (MI355X, ROCm 7.2, tools/dbg_bmm.py, each case in its own process), while M = 196, 1568, 3136 and 33320 of the same product
#          are correct; under the tuning loop [2, 4704, 1024] x [2, 1024, 64] and the one-sample in_proj / out_proj shapes fault too.
PAIR_GEMM = os.environ.get("DIFFMA_PAIR_GEMM", "own")
PAIR_OWN_MAX_ROWS = int(os.environ.get("DIFFMA_PAIR_OWN_MAX_ROWS", "3200"))


def _pair_use_bmm(t):
    return PAIR_GEMM == "bmm"


def _pair_use_own(rows, a, b, a_kmajor, b_kmajor, out_dtype=None):
    return PAIR_GEMM == "own" and rows <= PAIR_OWN_MAX_ROWS and hip_ops.gemm_supported(a, b, a_kmajor, b_kmajor, out_dtype)


def _own_pair(a, b, a_kmajor, b_kmajor, out, accumulate=False):
    """out[g] (+)= opA(a[g]) @ opB(b[g]) for g = 0, 1 in ONE launch (dm_gemm_n)."""
    with hip_ops.paired() as pr:
        for g in (0, 1):
            if g:
                pr.second()
            hip_ops.gemm(a[g], b[g], a_kmajor, b_kmajor, out=out[g], accumulate=accumulate)
    return out


def _bmm_untuned(x, y, out_dtype=None):
    tun = torch.cuda.tunable if x.is_cuda else None
    was = tun is not None and tun.is_enabled() and tun.tuning_is_enabled()
    if was:
        tun.tuning_enable(False)
    try:
        if out_dtype is None or out_dtype == x.dtype:
            return torch.bmm(x, y)
        try:
            return torch.bmm(x, y, out_dtype=out_dtype)
        except (RuntimeError, NotImplementedError, TypeError):
            return torch.bmm(x, y).to(out_dtype)
    finally:
        if was:
            tun.tuning_enable(True)


def _pair_matmul(x, y, out_dtype=None):
    """x [2, P, Q] @ y [2, Q, R] -> [2, P, R]: one batched GEMM, or one GEMM per mixer (see PAIR_GEMM)."""
    if _pair_use_bmm(x):
        return GemmChain.run(_bmm_untuned, x, y, out_dtype)
    if out_dtype == torch.float32 and x.dtype != torch.float32:
        return (_mm_f32(x[0], y[0]), _mm_f32(x[1], y[1]))      # indexable like a [2, ...] tensor: the caller only takes [0] and [1]
    out = torch.empty((2, x.shape[1], y.shape[2]), dtype=x.dtype, device=x.device)
    for g in (0, 1):
        GemmChain.run(torch.mm, x[g], y[g], out=out[g])
    return out


def _as_pair(a, b):
    """[2, ...] tensor holding a and b: the shared buffer when they already are its two halves (no copy), else a stack."""
    if (a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()):
        return torch.as_strided(a, (2,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))
    return torch.stack([a, b])


class _LinearPairFn(torch.autograd.Function):
    """(x0 @ W0^T, x1 @ W1^T) as ONE batched GEMM each way; bias-free (in_proj / out_proj of the mixers, block/mamba.py:259-261,315)."""

    @staticmethod
    def forward(ctx, x0, x1, W0, W1):
        dev = x0.device.type
        dt_ = torch.get_autocast_dtype(dev) if torch.is_autocast_enabled(dev) else x0.dtype
        xp = _as_pair(x0 if x0.dtype == dt_ else x0.to(dt_), x1 if x1.dtype == dt_ else x1.to(dt_))       # [2, B, L, K]
        Wst = stacked_pair(W0, W1, dt_)                                                                   # [2, N, K]
        ctx.save_for_backward(xp, Wst)
        ctx.meta = (x0.dtype, W0.dtype)
        K = xp.shape[-1]
        x2 = xp.view(2, -1, K)
        if _pair_use_own(x2.shape[1], x2[0], Wst[0], True, True):
            y = _own_pair(x2, Wst, True, True, torch.empty((2, x2.shape[1], Wst.shape[1]), dtype=dt_, device=xp.device))
        else:
            y = _pair_matmul(x2, Wst.transpose(1, 2))                                                      # [2, M, N]
        y = y.view(2, *x0.shape[:-1], Wst.shape[1])
        return y[0], y[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        xp, Wst = ctx.saved_tensors
        x_dt, w_dt = ctx.meta
        with torch.autocast(device_type=xp.device.type, enabled=False):
            N, K = Wst.shape[1], Wst.shape[2]
            dy = _as_pair(dy0.contiguous() if dy0.dtype == xp.dtype else dy0.contiguous().to(xp.dtype),
                          dy1.contiguous() if dy1.dtype == xp.dtype else dy1.contiguous().to(xp.dtype)).view(2, -1, N)
            x2 = xp.view(2, -1, K)
            M = dy.shape[1]
            if _pair_use_own(M, dy[0], Wst[0], True, False):
                dx = _own_pair(dy, Wst, True, False, torch.empty((2, M, K), dtype=dy.dtype, device=dy.device)).view(xp.shape)
            else:
                dx = _pair_matmul(dy, Wst).view(xp.shape)
            if dx.dtype != x_dt:
                dx = dx.to(x_dt)
            if _pair_use_own(M, dy[0], x2[0], False, False, torch.float32):
                dW = _own_pair(dy, x2, False, False, torch.empty((2, N, K), dtype=torch.float32, device=dy.device))
            elif _pair_use_bmm(dy):
                dW = _pair_matmul(dy.transpose(1, 2), x2, torch.float32)                                   # [2, N, K] fp32
            else:
                dW = (_tn_splitk(dy[0], x2[0]), _tn_splitk(dy[1], x2[1]))                                  # the unpaired path's products
            dW0, dW1 = (dW[0], dW[1]) if w_dt == torch.float32 else (dW[0].to(w_dt), dW[1].to(w_dt))
        return dx[0], dx[1], dW0, dW1


def linear_pair(x0, x1, W0, W1):
    """(F.linear(x0, W0), F.linear(x1, W1)) for the two mixers of a block: one batched GEMM forward, two backward."""
    return _LinearPairFn.apply(x0, x1, W0, W1)


class _SpiralSSMPairFn(torch.autograd.Function):
    """_SpiralSSMFn for the two mixers of a block at once (hoisted gate + hoisted softplus, 16-bit I/O, d_state 16, small launches):
    xz0, xz1 [B, L, 2*Din] -> (y0, y1) [B, L, Din], every kernel stage launched once for both."""

    @staticmethod
    def forward(ctx, xz0, xz1, idx0, idx1, grad_on, cw0, cw1, cb0, cb1, Wx0, Wx1, Wdt0, Wdt1, b0, b1, A0, A1, D0, D1):
        Bsz, L, D2 = xz0.shape
        Din = D2 // 2
        ndir = idx0.shape[0]
        R, N = Wdt0.shape[1], A0.shape[1]
        S, M = ndir * Bsz, ndir * Bsz * L
        dt_, dev = xz0.dtype, xz0.device
        need_grad = grad_on and any(ctx.needs_input_grad)
        xz, idx = (xz0, xz1), (idx0, idx1)
        cw, cb, bias, A, Dk = (cw0, cw1), (cb0, cb1), (b0, b1), (A0, A1), (D0, D1)
        Wx_st, Wdt_st = stacked_pair(Wx0, Wx1, dt_), stacked_pair(Wdt0, Wdt1, dt_)       # [2, R+2N, Din], [2, Din, R]
        xc = torch.empty((2, S, L, Din), dtype=dt_, device=dev)
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                hip_ops.gather_conv1d_fwd(xz[g][..., :Din], cw[g], cb[g], row_index=idx[g], ndir=ndir, silu=True, out=xc[g])
        # x_proj stays one product per mixer: as a batch-2 GEMM its [M, 1024] x [1024, 64] shape makes TunableOp's first-use tuning
        # run a library candidate that faults (MI355X, ROCm 7.2: memory access fault inside the tuning loop, tools/dbg_bmm.py);
        # the plain products below are the ones the unpaired path has always issued
        x_dbl = torch.empty((2, M, R + 2 * N), dtype=dt_, device=dev)
        xc2 = xc.view(2, M, Din)
        if _pair_use_own(M // 4, xc2[0], Wx_st[0], True, True):        # (a 64-column product: the own kernel holds up to ~4x the rows)
            _own_pair(xc2, Wx_st, True, True, x_dbl)
        else:
            for g in (0, 1):
                GemmChain.run(torch.mm, xc2[g], Wx_st[g].t(), out=x_dbl[g])
        delta = [None, None]
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                delta[g] = hip_ops.dtproj_softplus_fwd(x_dbl[g], Wdt_st[g], bias[g]).view(S, L, Din)
        ckpt = [hip_ops.alloc_scan_ckpt(S, L, N, Din, dt_, dev) if need_grad else None for _ in (0, 1)]
        xd3 = x_dbl.view(2, S, L, R + 2 * N)
        ydir = [None, None]
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                ydir[g] = hip_ops.scan_fwd(xc[g], delta[g], A[g], xd3[g][..., R:R + N], xd3[g][..., R + N:], Dk[g], None, bias[g], True,
                                           z_row_index=idx[g], out_row_index=idx[g], batch_per_dir=Bsz, ckpt=ckpt[g], delta_activated=True)
        y = torch.empty((2, Bsz, L, Din), dtype=dt_, device=dev)
        pre = torch.empty((2, Bsz, L, Din), dtype=dt_, device=dev) if need_grad else None
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                hip_ops.token_merge(ydir[g].view(ndir, Bsz, L, Din), gate=xz[g][..., Din:], pre_out=None if pre is None else pre[g], out=y[g])
        if need_grad:
            ctx.save_for_backward(xz0, xz1, idx0, idx1, cw0, cw1, cb0, cb1, b0, b1, A0, A1, D0, D1, Wx_st, Wdt_st, xc, x_dbl,
                                  delta[0], delta[1], ckpt[0], ckpt[1], pre)
            ctx.meta = (Wx0.dtype, Wdt0.dtype)
        return y[0], y[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        (xz0, xz1, idx0, idx1, cw0, cw1, cb0, cb1, b0, b1, A0, A1, D0, D1, Wx_st, Wdt_st, xc, x_dbl, dl0, dl1, ck0, ck1, pre) = ctx.saved_tensors
        wx_dt, wdt_dt = ctx.meta
        Bsz, L, D2 = xz0.shape
        Din = D2 // 2
        ndir = idx0.shape[0]
        R, N = Wdt_st.shape[2], A0.shape[1]
        S, M = ndir * Bsz, ndir * Bsz * L
        dt_, dev = xz0.dtype, xz0.device
        xz, idx, delta, ckpt = (xz0, xz1), (idx0, idx1), (dl0, dl1), (ck0, ck1)
        cw, cb, bias, A, Dk = (cw0, cw1), (cb0, cb1), (b0, b1), (A0, A1), (D0, D1)
        dy = [t.contiguous() if t.dtype == dt_ else t.contiguous().to(dt_) for t in (dy0, dy1)]
        dxz = torch.empty((2, Bsz, L, D2), dtype=dt_, device=dev)
        dx_dbl = torch.empty((2, M, R + 2 * N), dtype=dt_, device=dev)
        xd3 = x_dbl.view(2, S, L, R + 2 * N)
        gate_g = [None, None]
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                gate_g[g], _ = hip_ops.gate_bwd(dy[g], xz[g][..., Din:], pre[g], dz_out=dxz[g][..., Din:])
        du = torch.empty((2, S, L, Din), dtype=dt_, device=dev)
        res = [None, None]
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                res[g] = hip_ops.scan_bwd(xc[g], delta[g], A[g], xd3[g][..., R:R + N], xd3[g][..., R + N:], Dk[g], None, bias[g], gate_g[g], ckpt[g],
                                          True, z_row_index=idx[g], out_row_index=idx[g], batch_per_dir=Bsz, du_out=du[g],
                                          dbc_out=dx_dbl[g].view(S, L, R + 2 * N)[..., R:], delta_activated=True)
        ddelta = [res[g][1].view(M, Din) for g in (0, 1)]
        dWdt = [None, None]
        if hip_ops.dtproj_bwd_supported(ddelta[0], x_dbl[0], Wdt_st[0], dx_dbl[0]):
            with hip_ops.paired() as pr:
                for g in (0, 1):
                    if g:
                        pr.second()
                    dWdt[g] = hip_ops.dtproj_bwd(ddelta[g], x_dbl[g], Wdt_st[g], dx_dbl[g])
        else:
            for g in (0, 1):
                dx_dbl[g][:, :R] = GemmChain.run(torch.mm, ddelta[g], Wdt_st[g])
                dWdt[g] = _tn_splitk(ddelta[g], x_dbl[g][:, :R])
        xc2 = xc.view(2, M, Din)
        if _pair_use_own(M // 4, dx_dbl[0], xc2[0], False, False, torch.float32):
            dWx = _own_pair(dx_dbl, xc2, False, False, torch.empty((2, R + 2 * N, Din), dtype=torch.float32, device=dev))
        else:
            dWx = [_tn_splitk(dx_dbl[g], xc2[g]) for g in (0, 1)]                          # [R+2N, Din] fp32 each
        du2 = du.view(2, M, Din)
        if _pair_use_own(M // 4, dx_dbl[0], Wx_st[0], True, False):
            _own_pair(dx_dbl, Wx_st, True, False, du2, accumulate=True)                    # d x~ = du + dx_dbl @ Wx, in place, one launch
        else:
            for g in (0, 1):
                GemmChain.run(du2[g].addmm_, dx_dbl[g], Wx_st[g])
        dxc = du
        cres = [None, None]
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                cres[g] = hip_ops.gather_conv1d_bwd(xz[g][..., :Din], cw[g], cb[g], dxc[g], row_index=idx[g], ndir=ndir, silu=True)
        with hip_ops.paired() as pr:
            for g in (0, 1):
                if g:
                    pr.second()
                hip_ops.token_merge(cres[g][0].view(ndir, Bsz, L, Din), out=dxz[g][..., :Din])
        out = [dxz[0], dxz[1], None, None, None]
        per = lambda f: [f(0), f(1)]
        out += per(lambda g: cres[g][1].to(cw[g].dtype).reshape(cw[g].shape))
        out += per(lambda g: cres[g][2].to(cb[g].dtype) if cb[g] is not None else None)
        out += per(lambda g: dWx[g] if dWx[g].dtype == wx_dt else dWx[g].to(wx_dt))
        out += per(lambda g: dWdt[g] if dWdt[g].dtype == wdt_dt else dWdt[g].to(wdt_dt))
        out += per(lambda g: res[g][7].to(bias[g].dtype))
        out += per(lambda g: res[g][5].to(A[g].dtype))
        out += per(lambda g: res[g][6].to(Dk[g].dtype))
        return tuple(out)


def spiral_ssm_pair_supported(Bsz, L, dtype, mix0, mix1):
    """The call pattern the pair path is built for: two Mamba-1 mixers of equal shape on a ROCm device, 16-bit activations,
    d_state 16, a conv with bias, and a launch small enough that the unfused conv / chunk-parallel scans serve it."""
    if not (PAIR_MIXERS and HOIST_GATE and dtype in (torch.bfloat16, torch.float16)):
        return False
    idx0, idx1 = mix0.scan_index, mix1.scan_index
    if not idx0.is_cuda or idx0.shape != idx1.shape or idx0.shape[0] < 2 or idx0.shape[1] != L:
        return False
    if mix0.A_log.shape != mix1.A_log.shape or mix0.A_log.shape[1] != 16 or mix0.conv1d.bias is None or mix1.conv1d.bias is None:
        return False
    if mix0.conv1d.weight.shape != mix1.conv1d.weight.shape or mix0.x_proj.weight.shape != mix1.x_proj.weight.shape:
        return False
    if idx0.shape[0] * Bsz >= hip_ops.XPROJ_FUSED_MIN_SEQS:
        return False
    Din, R = mix0.dt_proj.weight.shape
    from . import _lib
    code = {torch.bfloat16: _lib.DM_BF16, torch.float16: _lib.DM_F16}[dtype]
    return bool(hip_ops.DTPROJ_FUSED and (R + 2 * 16) % 8 == 0 and _lib.load().dm_dtproj_softplus_supported(int(Din), int(R), code))


def spiral_ssm_pair(xz0, xz1, mix0, mix1, A0, A1):
    """The fused 3-direction operator (spiral_ssm) for the two mixers of a block in one set of launches."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _SpiralSSMPairFn.apply(xz0, xz1, mix0.scan_index, mix1.scan_index, torch.is_grad_enabled(),
                                      mix0.conv1d.weight, mix1.conv1d.weight, mix0.conv1d.bias, mix1.conv1d.bias,
                                      mix0.x_proj.weight, mix1.x_proj.weight, mix0.dt_proj.weight, mix1.dt_proj.weight,
                                      mix0.dt_proj.bias.float(), mix1.dt_proj.bias.float(), A0.float(), A1.float(),
                                      mix0.D.float(), mix1.D.float())


# ------------------------------------------------------------------------------------------------
# Token-major building blocks with autograd (used by the Mamba-2 mixer)
# ------------------------------------------------------------------------------------------------
class _GatherConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, scan_index):
        out = hip_ops.gather_conv1d_fwd(x, weight, bias, row_index=scan_index, ndir=scan_index.shape[0], silu=True)
        ctx.save_for_backward(x, weight, bias, scan_index)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, bias, scan_index = ctx.saved_tensors
        ndir = scan_index.shape[0]
        Bsz, L, Dm = x.shape
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        dx_slabs, dw, db = hip_ops.gather_conv1d_bwd(x, weight, bias, dout, row_index=scan_index, ndir=ndir, silu=True)
        dx = hip_ops.token_merge(dx_slabs.view(ndir, Bsz, L, Dm))
        return dx, dw.to(weight.dtype).reshape(weight.shape), (db.to(bias.dtype) if bias is not None else None), None


def gather_conv1d(x, conv_weight, conv_bias, scan_index):
    """x [B, L, C] token-major view -> SiLU(causal conv) of the token-gathered sequences, [ndir*B, L, C]."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _GatherConvFn.apply(x, conv_weight, conv_bias, scan_index)


class _IndexedScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, dt_bias, scan_index, Bsz, grad_on=True):
        S, L, Dm = u.shape
        N = A.shape[1]
        need_grad = grad_on and any(ctx.needs_input_grad[:8])
        ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, u.dtype, u.device) if need_grad else None
        A = A.contiguous()
        y = hip_ops.scan_fwd(u, delta, A, Bm, Cm, D, z, dt_bias, True, z_row_index=scan_index, out_row_index=scan_index,
                             batch_per_dir=Bsz, ckpt=ckpt)
        ctx.Bsz = Bsz
        ctx.save_for_backward(u, delta, A, Bm, Cm, D, z, dt_bias, scan_index, ckpt)
        return y.view(scan_index.shape[0], Bsz, L, Dm)

    @staticmethod
    def backward(ctx, dy):
        u, delta, A, Bm, Cm, D, z, dt_bias, scan_index, ckpt = ctx.saved_tensors
        ndir, Bsz = scan_index.shape[0], ctx.Bsz
        S, L, Dm = u.shape
        # dy is per direction here (the gated RMSNorm sits between the scan and the merge): the kernel gathers its rows through
        # the same table as the forward's scatter, indexed by sequence instead of by batch (DM_FLAG_DOUT_PER_SEQ); z is
        # gathered in the kernel and dz comes back per direction in token order, so the 3 slabs only need the merge sum.
        dy = dy.reshape(S, L, Dm).contiguous()
        if dy.dtype != u.dtype:
            dy = dy.to(u.dtype)
        du, ddelta, dzs, dB, dC, dA, dD, dbias = hip_ops.scan_bwd(u, delta, A, Bm, Cm, D, z, dt_bias, dy, ckpt, True,
                                                                  z_row_index=scan_index, out_row_index=scan_index,
                                                                  batch_per_dir=Bsz, dout_per_seq=True)
        dz = hip_ops.token_merge(dzs.view(ndir, Bsz, L, Dm))
        return (du, ddelta, dA.to(A.dtype), dB.to(Bm.dtype), dC.to(Cm.dtype), dD.to(D.dtype), dz.to(z.dtype), dbias.to(dt_bias.dtype),
                None, None, None)


class _SpiralSSDFn(torch.autograd.Function):
    """The Mamba-2 mixer between in_proj and out_proj as ONE autograd node: zxbcdt [B, L, 2*Din + 2N + H] (token-major
    in_proj output, column blocks [z | x | B | C | dt], block/mamba2.py:380-400) -> gated, RMS-normalised, merged y [B, L, Din].
    The backward assembles d(zxbcdt) in place from the kernels' outputs (no slice/cat/index autograd nodes, no zero fills)."""

    @staticmethod
    def forward(ctx, zxbcdt, conv_w, conv_b, dt_bias_h, A_h, D_h, norm_w, eps, scan_index, scan_index_inv, Din, N, grad_on=True):
        Bsz, L, _ = zxbcdt.shape
        H = A_h.shape[0]
        P = Din // H
        ndir = scan_index.shape[0]
        S, Cx = ndir * Bsz, Din + 2 * N
        z, xbc_in, dt_tok = zxbcdt[..., :Din], zxbcdt[..., Din:Din + Cx], zxbcdt[..., Din + Cx:]
        xBC = hip_ops.gather_conv1d_fwd(xbc_in, conv_w, conv_b, row_index=scan_index, ndir=ndir, silu=True)        # [S, L, Cx]
        x, Bm, Cm = xBC[..., :Din], xBC[..., Din:Din + N], xBC[..., Din + N:]
        need_grad = grad_on and any(ctx.needs_input_grad[:7])
        ctx.ssd = False
        if hip_ops.ssd_fwd_supported(xBC, L, P, N, views=(x, Bm, Cm, z)) and (not need_grad or hip_ops.ssd_bwd_supported(xBC, L, P, N)):
            # the matrix pipe (csrc/ssd.hip, ssd_bwd.hip): single-chunk SSD as dense tile products per (sequence, head), the per-head
            # dt read in the kernel through the gather table -- no [S, L, Din] delta tensor, no per-state recurrence, and nothing
            # but the operands saved for the backward (which recomputes the score tiles)
            ydir = hip_ops.ssd_fwd(x, Bm, Cm, dt_tok, z, A_h, D_h, dt_bias_h, z_row_index=scan_index, out_row_index=scan_index,
                                   batch_per_dir=Bsz)
            out, rstd = _SpiralSSDFn._norm_merge(ydir, norm_w, eps, ndir, Bsz, L, Din)
            if need_grad:
                ctx.ssd = True
                ctx.save_for_backward(zxbcdt, conv_w, conv_b, xBC, A_h, D_h, dt_bias_h, ydir, rstd, norm_w, scan_index)
                ctx.meta = (Din, N, H, P, eps, dt_bias_h.dtype, A_h.dtype, D_h.dtype)
            return out
        # dt is produced per token and per head: gather its rows per direction, broadcast head -> channels
        idx64 = scan_index.long()
        dt_g = torch.stack([dt_tok[:, idx64[k]] for k in range(ndir)])                                            # [ndir, B, L, H]
        delta = dt_g.reshape(S, L, H, 1).expand(S, L, H, P).reshape(S, L, Din)
        A = A_h.float().repeat_interleave(P)[:, None].expand(Din, N).contiguous()
        Dskip = D_h.float().repeat_interleave(P)
        dt_bias = dt_bias_h.float().repeat_interleave(P)
        ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Din, zxbcdt.dtype, zxbcdt.device) if need_grad else None
        ydir = hip_ops.scan_fwd(x, delta, A, Bm, Cm, Dskip, z, dt_bias, True, z_row_index=scan_index, out_row_index=scan_index,
                                batch_per_dir=Bsz, ckpt=ckpt, a_shared=True)     # token order, gated; one decay per head
        out, rstd = _SpiralSSDFn._norm_merge(ydir, norm_w, eps, ndir, Bsz, L, Din)
        ctx.save_for_backward(zxbcdt, conv_w, conv_b, xBC, delta, A, Dskip, dt_bias, ckpt, ydir, rstd, norm_w, scan_index, scan_index_inv)
        ctx.meta = (Din, N, H, P, eps, dt_bias_h.dtype, A_h.dtype, D_h.dtype)
        return out

    @staticmethod
    def _norm_merge(ydir, norm_w, eps, ndir, Bsz, L, Din):
        """Gated RMSNorm of every direction + merge; norm_w None (the operator called with rmsnorm_weight=None, one direction):
        the gated scan output itself."""
        if norm_w is None:
            if ndir != 1:
                raise NotImplementedError("the un-normalised Mamba-2 core is wired for one direction (the reference-facing operator)")
            return ydir.view(Bsz, L, Din), None
        return hip_ops.rmsnorm_merge_fwd(ydir.view(ndir, Bsz, L, Din), norm_w, eps)

    @staticmethod
    def _norm_merge_bwd(ydir, norm_w, eps, rstd, dout, ndir, Bsz, L, Din):
        if norm_w is None:
            return dout.contiguous().view(1, Bsz, L, Din), None
        return hip_ops.rmsnorm_merge_bwd(ydir.view(ndir, Bsz, L, Din), norm_w, eps, rstd, dout)

    @staticmethod
    def backward(ctx, dout):
        if ctx.ssd:
            return _SpiralSSDFn._backward_matrix_pipe(ctx, dout)
        zxbcdt, conv_w, conv_b, xBC, delta, A, Dskip, dt_bias, ckpt, ydir, rstd, norm_w, scan_index, scan_index_inv = ctx.saved_tensors
        Din, N, H, P, eps, bias_dt, A_dt, D_dt = ctx.meta
        Bsz, L, _ = zxbcdt.shape
        ndir = scan_index.shape[0]
        S, Cx = ndir * Bsz, Din + 2 * N
        dt_ = zxbcdt.dtype
        if dout.dtype != dt_:
            dout = dout.to(dt_)
        dyd, dnorm_w = _SpiralSSDFn._norm_merge_bwd(ydir, norm_w, eps, rstd, dout, ndir, Bsz, L, Din)                # [ndir, B, L, Din]
        dxBC = torch.empty((S, L, Cx), dtype=dt_, device=zxbcdt.device)
        x, Bm, Cm = xBC[..., :Din], xBC[..., Din:Din + N], xBC[..., Din + N:]
        _, ddelta, dzs, _, _, dA, dD, dbias = hip_ops.scan_bwd(
            x, delta, A, Bm, Cm, Dskip, zxbcdt[..., :Din], dt_bias, dyd.view(S, L, Din), ckpt, True, z_row_index=scan_index,
            out_row_index=scan_index, batch_per_dir=Bsz, dout_per_seq=True, du_out=dxBC[..., :Din], a_shared=True,
            dbc_out=dxBC[..., Din:])                                             # dB | dC land in their xBC columns
        dx_slabs, dconv_w, dconv_b = hip_ops.gather_conv1d_bwd(zxbcdt[..., Din:Din + Cx], conv_w, conv_b, dxBC, row_index=scan_index,
                                                               ndir=ndir, silu=True)                              # token order
        dzx = torch.empty_like(zxbcdt)
        hip_ops.token_merge(dzs.view(ndir, Bsz, L, Din), out=dzx[..., :Din])
        hip_ops.token_merge(dx_slabs.view(ndir, Bsz, L, Cx), out=dzx[..., Din:Din + Cx])
        # d(dt): sum the head's channels, put every direction back in token order (adjoint of the row gather), add the directions
        # (a matrix-vector product: the generic inner-dimension reduction kernel takes 150 us for this shape, the GEMV 20)
        ddt = torch.mv(ddelta.view(S * L * H, P), torch.ones(P, dtype=dt_, device=zxbcdt.device)).view(ndir, Bsz, L, H)
        inv64 = scan_index_inv.long()
        ddt_tok = ddt[0][:, inv64[0]].float()
        for k in range(1, ndir):
            ddt_tok = ddt_tok + ddt[k][:, inv64[k]].float()
        dzx[..., Din + Cx:].copy_(ddt_tok)
        dA_h = dA.view(H, P * N).sum(-1)
        dD_h = dD.view(H, P).sum(-1)
        dbias_h = dbias.view(H, P).sum(-1)
        return (dzx, dconv_w.to(conv_w.dtype).reshape(conv_w.shape), dconv_b.to(conv_b.dtype) if conv_b is not None else None,
                dbias_h.to(bias_dt), dA_h.to(A_dt), dD_h.to(D_dt), None if dnorm_w is None else dnorm_w.to(norm_w.dtype), None, None, None, None, None, None)


    @staticmethod
    def _backward_matrix_pipe(ctx, dout):
        zxbcdt, conv_w, conv_b, xBC, A_h, D_h, dt_bias_h, ydir, rstd, norm_w, scan_index = ctx.saved_tensors
        Din, N, H, P, eps, bias_dt, A_dt, D_dt = ctx.meta
        Bsz, L, _ = zxbcdt.shape
        ndir = scan_index.shape[0]
        S, Cx = ndir * Bsz, Din + 2 * N
        dt_ = zxbcdt.dtype
        if dout.dtype != dt_:
            dout = dout.to(dt_)
        dyd, dnorm_w = _SpiralSSDFn._norm_merge_bwd(ydir, norm_w, eps, rstd, dout, ndir, Bsz, L, Din)                # [ndir, B, L, Din]
        dxBC = torch.empty((S, L, Cx), dtype=dt_, device=zxbcdt.device)
        x, Bm, Cm = xBC[..., :Din], xBC[..., Din:Din + N], xBC[..., Din + N:]
        _, dzs, dbc, ddt, dad = hip_ops.ssd_bwd(
            x, Bm, Cm, zxbcdt[..., Din + Cx:], zxbcdt[..., :Din], dyd.view(S, L, Din), A_h, D_h, dt_bias_h, z_row_index=scan_index,
            out_row_index=scan_index, batch_per_dir=Bsz, dx_out=dxBC[..., :Din])
        dxBC[..., Din:].copy_(dbc)                                                    # dB | dC (summed over the heads) in their xBC columns
        dx_slabs, dconv_w, dconv_b = hip_ops.gather_conv1d_bwd(zxbcdt[..., Din:Din + Cx], conv_w, conv_b, dxBC, row_index=scan_index,
                                                               ndir=ndir, silu=True)                              # token order
        dzx = torch.empty_like(zxbcdt)
        hip_ops.token_merge(dzs.view(ndir, Bsz, L, Din), out=dzx[..., :Din])
        hip_ops.token_merge(dx_slabs.view(ndir, Bsz, L, Cx), out=dzx[..., Din:Din + Cx])
        ddt4 = ddt.view(ndir, Bsz, L, H)                                              # already in token order, raw-dt gradient: add the directions
        ddt_tok = ddt4[0] if ndir == 1 else ddt4.sum(0)
        dzx[..., Din + Cx:].copy_(ddt_tok)
        return (dzx, dconv_w.to(conv_w.dtype).reshape(conv_w.shape), dconv_b.to(conv_b.dtype) if conv_b is not None else None,
                dad[2].to(bias_dt), dad[0].to(A_dt), dad[1].to(D_dt), None if dnorm_w is None else dnorm_w.to(norm_w.dtype), None, None, None, None, None, None)


def spiral_ssd(zxbcdt, conv_w, conv_b, dt_bias, A, D, norm_w, eps, scan_index, scan_index_inv, d_inner, d_state):
    """Fused core of the Mamba-2 'spiral' mixer (see _SpiralSSDFn).  A = -exp(A_log) [H], dt_bias [H], D [H]."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _SpiralSSDFn.apply(zxbcdt, conv_w, conv_b, dt_bias, A, D, norm_w, eps, scan_index, scan_index_inv, d_inner, d_state,
                                  torch.is_grad_enabled())


def indexed_scan(u, delta, A, Bm, Cm, D, z, dt_bias, scan_index, Bsz):
    """Selective scan of ndir*B token-gathered sequences with the z gather and the inverse (merge) reindex folded in.
    u, delta: [ndir*B, L, Dm]; Bm, Cm: [ndir*B, L, N] views; z: [B, L, Dm] (token order).  Returns the gated output
    [ndir, B, L, Dm] with every direction already back in TOKEN order."""
    with torch.autocast(device_type="cuda", enabled=False):
        if delta.dtype != u.dtype:
            delta = delta.to(u.dtype)
        if z is None:
            raise NotImplementedError("norm_before_gate=True (ungated scan) is not wired; DiffMa uses norm_before_gate=False")
        return _IndexedScanFn.apply(u, delta.contiguous(), A.float(), Bm, Cm, D.float(), z, dt_bias.float(), scan_index, Bsz,
                                    torch.is_grad_enabled())


class _RmsMergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, slabs, weight, eps):
        slabs = slabs.contiguous()
        out, rstd = hip_ops.rmsnorm_merge_fwd(slabs, weight, eps)
        ctx.save_for_backward(slabs, weight, rstd)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, dout):
        slabs, weight, rstd = ctx.saved_tensors
        if dout.dtype != slabs.dtype:
            dout = dout.to(slabs.dtype)
        dy, dw = hip_ops.rmsnorm_merge_bwd(slabs, weight, ctx.eps, rstd, dout)
        return dy, dw.to(weight.dtype), None


def rmsnorm_merge(slabs, weight, eps):
    """[K, B, L, C] gated scan outputs (token order) -> weight * sum_k RMSNorm(slab_k): the Mamba-2 gated RMSNorm
    (norm_before_gate = False, block/mamba2.py:349) fused with the 3-way CrossMerge."""
    with torch.autocast(device_type="cuda", enabled=False):
        return _RmsMergeFn.apply(slabs, weight, eps)


def mamba_split_conv1d_scan_combined(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size, initial_states=None,
                                     seq_idx=None, dt_limit=(0.0, float("inf")), return_final_states=False, activation="silu",
                                     rmsnorm_weight=None, rmsnorm_eps=1e-6, outproj_weight=None, outproj_bias=None, headdim=None,
                                     ngroups=1, norm_before_gate=True):
    """Drop-in for mamba_ssm's Mamba-2 fused operator with the keyword usage of block/mamba2.py:392-410.

    zxbcdt: (B, L, 2*dim + 2*ngroups*N + H), columns [z | x | B | C | dt].  Single chunk (chunk_size >= L, the only case
    DiffMa reaches: chunk_size 256, L <= 196), scalar D per head, ngroups 1.  Returns (B, L, d_model)."""
    if initial_states is not None or seq_idx is not None or return_final_states or ngroups != 1 or headdim is None:
        raise NotImplementedError("only the call pattern of block/mamba2.py:392-410 is built (ngroups=1, scalar D, no states)")
    if activation not in ("silu", "swish") or dt_limit != (0.0, float("inf")):
        raise NotImplementedError("activation must be silu and dt_limit unbounded")
    Bsz, L, _ = zxbcdt.shape
    if chunk_size < L:
        raise NotImplementedError("multi-chunk SSD (chunk_size < seqlen) is never reached by DiffMa")
    if rmsnorm_weight is not None and norm_before_gate:
        raise NotImplementedError("norm_before_gate=True (norm, then gate) is not wired; DiffMa uses norm_before_gate=False (block/mamba2.py:349)")
    H = D.shape[0]
    dim = H * headdim
    N = (zxbcdt.shape[-1] - 2 * dim - H) // 2
    if zxbcdt.stride(-1) != 1:
        zxbcdt = zxbcdt.contiguous()
    # ONE autograd node, the same one the Mamba2 module runs with three directions: conv (K3), the single-chunk SSD core on the
    # matrix pipe (dm_ssd_fwd / dm_ssd_bwd) when the shape allows it -- 16-bit I/O, headdim 64, d_state 16, 16-byte aligned rows --
    # else the A-shared selective scan, then the gated RMSNorm (dm_rmsnorm_merge_*, one slab) when rmsnorm_weight is given.
    ident = _const("ident", L, zxbcdt.device)
    y = spiral_ssd(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, rmsnorm_weight, rmsnorm_eps, ident, ident, dim, N)
    if outproj_weight is not None:
        y = F.linear(y, outproj_weight.to(y.dtype), outproj_bias)
    return y


def spiral_ssm(xz, conv_w, conv_b, x_proj_w, dt_proj_w, dt_proj_b, A, Dskip, scan_index, out_index=None, merge=True):
    """Fused CrossScan -> ndir x (conv1d+SiLU, x_proj, dt_proj, selective scan) -> CrossMerge (pre out_proj).

    xz: [B, L, 2*Din] token-major (the in_proj output); A: [Din, N] fp32 (= -exp(A_log));
    scan_index: int32 [ndir, L] device tensor (3 spiral directions for DiffMa; 1 / 2 / 4 for the baseline blocks).
    out_index / merge: see _SpiralSSMFn.forward.  Returns y [B, L, Din] (or [ndir, B, L, Din]); the caller applies out_proj.
    """
    with torch.autocast(device_type="cuda", enabled=False):
        return _SpiralSSMFn.apply(xz, conv_w, conv_b, x_proj_w, dt_proj_w, dt_proj_b.float(), A.float(), Dskip.float(),
                                  scan_index, torch.is_grad_enabled(), out_index, merge)


_CONSTS = {}


def _const(kind, n, device):
    """Identity row table / zero vector of the reference-facing operators, cached per (kind, size, device): the per-call
    torch.arange / torch.zeros were two launches of every mamba_inner_fn call (three calls per mixer on route A)."""
    key = (kind, int(n), str(device))
    t = _CONSTS.get(key)
    if t is None:
        # created OUTSIDE inference mode (an inference tensor cannot be saved for a later backward); not cached while a stream is
        # capturing (the tensor would live in that graph's private pool) -- ADVICE r4
        with torch.inference_mode(False):
            t = torch.arange(n, device=device, dtype=torch.int32)[None] if kind == "ident" else torch.zeros(n, device=device)
        if not (torch.device(device).type == "cuda" and torch.cuda.is_current_stream_capturing()):
            _CONSTS[key] = t
    return t


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                   out_proj_bias, A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                   C_proj_bias=None, delta_softplus=True):
    """Drop-in for mamba_ssm's mamba_inner_fn with the positional usage of block/mamba.py:346.

    xz: (B, 2*Din, L) (any strides; CrossScan hands out strided slices).  Returns (B, L, d_model).
    """
    if B is not None or C is not None or B_proj_bias is not None or C_proj_bias is not None:
        raise NotImplementedError("constant B/C and B/C projection biases are never used by DiffMa")
    if not delta_softplus:
        raise NotImplementedError("mamba_inner_fn is only defined with delta_softplus=True upstream")
    # [B, L, 2Din]: a view when xz is a transposed token-major buffer, one dm_repack pass when it is the reference's L-contiguous slice
    xzt = xz.transpose(1, 2) if xz.stride(1) == 1 else _RepackFn.apply(xz)
    ident = _const("ident", xzt.shape[1], xz.device)                              # made once per (length, device), not per call
    bias = delta_bias if delta_bias is not None else _const("zeros", xzt.shape[2] // 2, xz.device)
    Dsk = D if D is not None else _const("zeros", xzt.shape[2] // 2, xz.device)
    y = spiral_ssm(xzt, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, bias, A, Dsk, ident)
    # out_proj the way the native mixer runs it: K11 / K12 / the tuned library by size, the input gradient as an NT product, the
    # weight gradient split over the B L rows it contracts (F.linear's autograd hands those to untuned single-pass GEMMs)
    return linear_splitk(y, out_proj_weight, out_proj_bias)
