"""Thin, allocation-explicit wrappers over the C ABI (include/diffma_hip.h).

Everything here works on TOKEN-MAJOR tensors ([seq, L, D], channel stride 1) -- the layout the
gfx950 kernels are written for -- and on raw device pointers + the current torch stream.  The
reference-facing operator names live in selective_scan_interface.py.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import (DM_BF16, DM_F16, DM_F32, DM_FLAG_A_SHARED, DM_FLAG_DELTA_ACTIVATED, DM_FLAG_DX_MERGED, DM_FLAG_PARTIAL_COMPACT, DM_FLAG_OUT_ACCUMULATE, DM_FLAG_DELTA_SOFTPLUS, DM_FLAG_DOUT_PER_SEQ, DM_FLAG_SCAN_CHUNKED, DM_FLAG_SCAN_SEQUENTIAL, DM_FLAG_SILU, dm_blend_args, dm_gate_head_args, dm_dtproj_bwd_args, dm_conv_bwd_args, dm_rmsnorm_merge_args, dm_colsum_args, dm_sum_partials_args,
                   dm_conv_fwd_args, dm_conv_xproj_bwd_args, dm_conv_xproj_fwd_args, dm_diffusion_step_args, dm_ln_mod_args, dm_ssd_bwd_args, dm_ssd_fwd_args, dm_merge_args, dm_gate_bwd_args, dm_dtproj_args, dm_scan_bwd_args, dm_scan_fwd_args, dm_gemm_args, dm_repack_args, dm_training_loss_args)

_DT = {torch.float32: DM_F32, torch.bfloat16: DM_BF16, torch.float16: DM_F16}
SCAN_CKPT_EVERY = 4          # forward checkpoint spacing = backward sub-chunk length (csrc/scan_bwd_impl.h BWD_SUB)


def union_length(intervals):
    """Total length of the union of (start, end) intervals (overlapping launches of one kernel count once)."""
    total, cur_s, cur_e = 0.0, None, None
    for s0, e0 in sorted(intervals):
        if cur_e is None or s0 > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


class KernelTimer:
    """Optional live timing of the C-ABI launches with events recorded on the launch stream (the torch
    current stream).  bench.py installs one around its timed region; no host sync until `summary()`."""

    def __init__(self):
        self.records = {}      # name -> list of (start_event, end_event, algorithmic_bytes, design_bytes)

    def launch(self, name, nbytes, fn, design_bytes=None, flops=0):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.records.setdefault(name, []).append((e0, e1, nbytes, nbytes if design_bytes is None else design_bytes, flops))

    def summary(self):
        """Per kernel: launches, avg_us / total_ms (per-launch durations, what a kernel trace reports), bytes_per_launch, and
        busy_ms = the length of the UNION of the launch intervals.  The two mixers of a block run on two streams, so two
        launches of the same kernel can share the GPU: each then takes about twice as long, while total_ms / busy_ms (the
        mean number of concurrent launches) tells how many were sharing it."""
        torch.cuda.synchronize()
        out = {}
        base = None
        for recs in self.records.values():
            if recs and base is None:
                base = recs[0][0]
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            nb = [r[2] for r in recs]
            db = [r[3] for r in recs]
            busy = union_length((base.elapsed_time(r[0]), base.elapsed_time(r[0]) + d) for r, d in zip(recs, ms))
            out[name] = dict(launches=len(recs), avg_us=1e3 * sum(ms) / len(ms), total_ms=sum(ms),
                             bytes_per_launch=sum(nb) / len(nb), design_bytes_per_launch=sum(db) / len(db), busy_ms=busy,
                             flops_per_launch=sum(r[4] for r in recs) / len(recs))
        return out


_TIMER = None


def set_timer(timer):
    """Install (or remove with None) a KernelTimer; returns the previous one."""
    global _TIMER
    prev, _TIMER = _TIMER, timer
    return prev


class paired:
    """Context manager that makes the two mixers of a DiffMa block share their kernel launches (reference block/mamba_block.py:107-108
    runs them one after the other; at the reference's own batch -- config/brain.yaml, one sample per GPU -- a step is bound by the
    number of launches).  Inside the block the C-ABI launches of this module are QUEUED instead of issued; the caller runs the same
    wrapper calls first for mixer 0, then for mixer 1; at exit launch i of mixer 0 and launch i of mixer 1 go out as ONE call of the
    kernel's `_n` entry point (include/diffma_hip.h, ABI 25), which puts congruent launches into one grid (blockIdx.z picks the
    argument struct) and falls back to two launches otherwise.  Results are bit-identical to the unpaired calls.
    Rules for the code inside: only wrapper calls, allocations and views -- no torch arithmetic on their outputs (they have not
    been computed yet); a wrapper that has to fall back to torch arithmetic calls `_pair_flush()` first, which issues everything
    queued so far one by one and turns the rest of the block into immediate launches.  Every tensor whose pointer went into a
    queued argument struct is kept alive until the launch (`_ptr`)."""

    def __init__(self, enabled=True):
        self.enabled = enabled
        self.queue, self.keep, self.broken, self.mark = [], [], False, None

    def __enter__(self):
        if self.enabled:
            if _pair() is not None:
                raise RuntimeError("hip_ops.paired() does not nest")
            _TLS.pair = self
        return self

    def second(self):
        """Call between the first and the second mixer's wrapper calls."""
        self.mark = len(self.queue)

    def __exit__(self, et, ev, tb):
        if not self.enabled:
            return False
        _TLS.pair = None
        if et is None:
            self.flush(pairwise=True)
        self.queue, self.keep = [], []
        return False

    def flush(self, pairwise=False):
        q, self.queue = self.queue, []
        k = self.mark
        if pairwise and not self.broken and k is not None and len(q) == 2 * k and all(q[i][0] == q[i + k][0] for i in range(k)):
            for i in range(k):
                name, a0, tensor, nbytes, design, flops = q[i]
                _issue(name, [a0, q[i + k][1]], tensor, nbytes + q[i + k][3], None if design is None else design + (q[i + k][4] or 0),
                       flops + q[i + k][5])
        else:
            for name, a, tensor, nbytes, design, flops in q:
                _issue(name, [a], tensor, nbytes, design, flops)
        self.keep = []
        self.mark = None


import threading

_TLS = threading.local()          # the queue belongs to the thread that opened it: the backward runs on autograd's worker thread


def _pair():
    return getattr(_TLS, "pair", None)


def _pair_flush():
    """A wrapper is about to do torch arithmetic on launch outputs: issue what is queued (unpaired) and stop queueing."""
    pr = _pair()
    if pr is not None:
        pr.flush()
        pr.broken = True


_DEBUG_SYNC = os.environ.get("DIFFMA_DEBUG_SYNC", "0") == "1"      # developer aid: synchronise and name every C-ABI launch


def _issue(name, arg_list, tensor, nbytes, design_bytes, flops=0):
    if _DEBUG_SYNC:
        print(f"[hip_ops] {name} x{len(arg_list)}", flush=True)
    with torch.cuda.device(tensor.device):
        if len(arg_list) == 1:
            fn = lambda: _lib.call(name, arg_list[0], _stream(tensor))
        elif _lib.has_n(name):
            fn = lambda: _lib.call_n(name, arg_list, _stream(tensor))
        else:
            def fn():
                for a in arg_list:
                    _lib.call(name, a, _stream(tensor))
        if _TIMER is None:
            fn()
        else:
            _TIMER.launch(name, nbytes, fn, design_bytes, flops)
        if _DEBUG_SYNC:
            torch.cuda.synchronize()


def _launch(name, args, tensor, nbytes, design_bytes=None, flops=0):
    """nbytes: ALGORITHMIC bytes of the launch (what any implementation of the operator must move);
    design_bytes: the bytes THIS implementation moves by design (algorithmic + checkpoints + partial rows), if different;
    flops: for the matrix-pipe kernels whose roof is the MFMA peak (dm_gemm)."""
    pr = _pair()
    if pr is not None and not pr.broken:
        pr.queue.append((name, args, tensor, nbytes, design_bytes, flops))
        return
    _issue(name, [args], tensor, nbytes, design_bytes, flops)


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; the kernels take fp32, bf16 or fp16") from None


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "DiffMa HIP operators need tensors on a ROCm device; there is no CPU implementation in "
                "the product path (the CPU oracle under oracle/ is test infrastructure only)")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    if t is None:
        return 0
    pr = _pair()
    if pr is not None:
        pr.keep.append(t)              # the launch is deferred: the tensor must outlive the wrapper call that named it
    return t.data_ptr()


def _f32c(t):
    """fp32 contiguous view/copy of a small parameter tensor."""
    if t is None:
        return None
    return t.detach().to(torch.float32).contiguous()


def scan_nchunk(L: int, every: int = SCAN_CKPT_EVERY) -> int:
    return (L + every - 1) // every


def alloc_scan_ckpt(S: int, L: int, N: int, Dm: int, io_dtype, device):
    """Checkpoint buffer of the training forward (include/diffma_hip.h): bf16 I/O stores pairs of bf16 states in
    one 32-bit word ([S, chunk, N/8 * Dm, 4] int32), every other I/O dtype stores fp32 ([S, chunk, N, Dm])."""
    if io_dtype == torch.bfloat16:
        return torch.empty((S, scan_nchunk(L), (N // 8) * Dm, 4), dtype=torch.int32, device=device)
    return torch.empty((S, scan_nchunk(L), N, Dm), dtype=torch.float32, device=device)


def _ckpt_dtype_code(ckpt):
    if ckpt is None:
        return DM_F32
    return DM_BF16 if ckpt.dtype == torch.int32 else DM_F32


def scan_fwd_algorithmic_bytes(S, Dm, L, N, es, es_bc, has_z=True):
    """SURVEY.md 8(d) / BASELINE.md section 3: selective_scan_fn forward reads u, delta, z and writes out (4*s B per element),
    reads the shared B, C rows and the constants A, D, delta_bias.  Nothing implementation-specific is counted."""
    return (4 if has_z else 3) * S * Dm * L * es + 2 * S * N * L * es_bc + 4 * Dm * N + 8 * Dm


def scan_bwd_algorithmic_bytes(S, Dm, L, N, es, has_z=True):
    """the backward reads u, delta, z, dout and writes du, ddelta, dz (7*s B per element) plus fp32 dB, dC
    (2*S*N*L*4).  Checkpoint reads, per-workgroup partial rows and the dA/dD/dbias partials are design traffic, not counted."""
    return (7 if has_z else 5) * S * Dm * L * es + 2 * S * N * L * 4


def _variant_flag(variant):
    """variant: None = the library chooses by launch size; "sequential" / "chunked" force a kernel family (tests, A/B runs)."""
    if variant is None:
        return 0
    return {"sequential": DM_FLAG_SCAN_SEQUENTIAL, "chunked": DM_FLAG_SCAN_CHUNKED}[variant]


def scan_fwd(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, delta_softplus=True, *,
             z_row_index=None, out_row_index=None, batch_per_dir=0, out=None, ckpt=None,
             ckpt_every=SCAN_CKPT_EVERY, last_state=None, ngroups=1, a_shared=False, variant=None, acc_dirs=False,
             delta_activated=False):
    """u, delta: [S, L, Dm] token-major (last stride 1).  Bm, Cm: [S, L, G*N] views (state stride 1).
    z: [S or S/ndir, Lz, Dm] or None.  A: [Dm, N] fp32.  Returns out [S, L, Dm] (allocated if None).
    acc_dirs: the ndir directions are accumulated into ONE token-order buffer [batch_per_dir, L, Dm], which is returned: one
    launch per direction, the first stores, the others add (DM_FLAG_OUT_ACCUMULATE) -- CrossMerge folded into the scan.
    delta_activated: delta already holds softplus(raw + bias) (dtproj_softplus); delta_bias / delta_softplus are then ignored."""
    _require_gpu(u, delta, A, Bm, Cm, z)
    S, L, Dm = u.shape
    N = A.shape[1]
    if acc_dirs:
        Bd = batch_per_dir
        ndir = S // Bd
        if out is None:
            out = torch.empty((Bd, L, Dm), dtype=u.dtype, device=u.device)
        for k in range(ndir):
            sl = slice(k * Bd, (k + 1) * Bd)
            _scan_fwd_launch(u[sl], delta[sl], A, Bm[sl], Cm[sl], D, z, delta_bias, delta_softplus, z_row_index[k:k + 1],
                             out_row_index[k:k + 1], 0, out, None if ckpt is None else ckpt[sl], ckpt_every, None, ngroups, False,
                             "sequential", k > 0)
        return out
    if out is None:
        out = torch.empty((S, L, Dm), dtype=u.dtype, device=u.device)
    return _scan_fwd_launch(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, z_row_index, out_row_index, batch_per_dir, out, ckpt,
                            ckpt_every, last_state, ngroups, a_shared, variant, False, delta_activated)


def _scan_fwd_launch(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, z_row_index, out_row_index, batch_per_dir, out, ckpt,
                     ckpt_every, last_state, ngroups, a_shared, variant, accumulate, delta_activated=False):
    S, L, Dm = u.shape
    N = A.shape[1]
    A = _f32c(A)
    D = _f32c(D)
    delta_bias = _f32c(delta_bias)
    a = dm_scan_fwd_args()
    a.nseq, a.dim, a.seqlen, a.dstate = S, Dm, L, N
    a.ngroups = ngroups
    a.batch_per_dir = batch_per_dir
    a.io_dtype = dtype_code(u)
    a.bc_dtype = dtype_code(Bm)
    a.flags = ((DM_FLAG_DELTA_SOFTPLUS if delta_softplus else 0) | (DM_FLAG_A_SHARED if a_shared else 0) | _variant_flag(variant)
               | (DM_FLAG_OUT_ACCUMULATE if accumulate else 0))
    if delta_activated:
        a.flags = (a.flags & ~DM_FLAG_DELTA_SOFTPLUS) | DM_FLAG_DELTA_ACTIVATED
    a.ckpt_every = ckpt_every
    a.ckpt_dtype = _ckpt_dtype_code(ckpt)
    a.u, a.delta, a.z, a.out = _ptr(u), _ptr(delta), _ptr(z), _ptr(out)
    a.B, a.C, a.A, a.D, a.delta_bias = _ptr(Bm), _ptr(Cm), _ptr(A), _ptr(D), _ptr(delta_bias)
    a.z_row_index, a.out_row_index = _ptr(z_row_index), _ptr(out_row_index)
    a.ckpt, a.last_state = _ptr(ckpt), _ptr(last_state)
    a.u_ss, a.u_sl, a.u_sd = u.stride()
    a.dt_ss, a.dt_sl, a.dt_sd = delta.stride()
    if z is not None:
        a.z_ss, a.z_sl, a.z_sd = z.stride()
    a.o_ss, a.o_sl, a.o_sd = out.stride()
    a.B_ss, a.B_sl, a.B_sn = Bm.stride()
    a.C_ss, a.C_sl, a.C_sn = Cm.stride()
    a.B_sg = a.C_sg = N
    nbytes = scan_fwd_algorithmic_bytes(S, Dm, L, N, u.element_size(), Bm.element_size(), z is not None)
    design = nbytes
    if accumulate:                               # the running sum is read back once per accumulating launch
        design += S * Dm * L * u.element_size()
    if ckpt is not None:
        design += scan_nchunk(L, ckpt_every) * S * ckpt.shape[2] * ckpt.shape[3] * 4     # slots 1.. + slot 0 (the final state)
    _launch("dm_selective_scan_fwd", a, u, nbytes, design)
    return out


def scan_bwd(u, delta, A, Bm, Cm, D, z, delta_bias, dout, ckpt, delta_softplus=True, *,
             z_row_index=None, out_row_index=None, batch_per_dir=0, ckpt_every=SCAN_CKPT_EVERY,
             ngroups=1, dz_out=None, dout_per_seq=False, du_out=None, a_shared=False, dbc_out=None, variant=None,
             delta_activated=False):
    """Reverse-time pass.  Returns (du, ddelta, dz, dB, dC, dA, dD, dbias) with parameter gradients
    already reduced over sequences.  dz is [S, Lz, Dm] in the z buffer's row order (token order when
    z_row_index is given).  delta_activated: delta holds softplus(raw + bias) already (DM_FLAG_DELTA_ACTIVATED); ddelta and dbias
    are still the gradients of the raw value / of the bias."""
    _require_gpu(u, delta, A, Bm, Cm, z, dout, ckpt)
    S, L, Dm = u.shape
    N = A.shape[1]
    flags = ((DM_FLAG_DELTA_SOFTPLUS if delta_softplus else 0) | (DM_FLAG_DOUT_PER_SEQ if dout_per_seq else 0)
             | (DM_FLAG_A_SHARED if a_shared else 0) | _variant_flag(variant))
    if delta_activated:
        flags = (flags & ~DM_FLAG_DELTA_SOFTPLUS) | DM_FLAG_DELTA_ACTIVATED
    gc = _lib.load().dm_scan_bwd_launch_group_channels(S, Dm, L, N, flags)   # 256 (sequential kernel) or 64 (chunk-parallel, small launches)
    if gc <= 0:
        raise _lib.DiffmaHipError(f"selective-scan backward is not built for d_state={N}")
    nw = (Dm + gc - 1) // gc
    dev = u.device
    A32, D32, b32 = _f32c(A), _f32c(D), _f32c(delta_bias)
    du = du_out if du_out is not None else torch.empty_like(u)      # du_out: a [S, L, Dm] view with channel stride 1
    ddelta = torch.empty((S, L, Dm), dtype=u.dtype, device=dev)
    dz = None
    if z is not None:
        dz = dz_out if dz_out is not None else torch.empty((S, L, Dm), dtype=u.dtype, device=dev)
    dBC = torch.empty((S, L, nw, 2 * N), dtype=torch.float32, device=dev)
    # per-sequence partial rows of dA | dD | dbias as column blocks of ONE buffer: one column sum instead of three
    want_dD, want_db = D is not None, (delta_bias is not None or delta_activated)
    pcols = Dm * N + Dm * (int(want_dD) + int(want_db))
    part = torch.empty((S, pcols), dtype=torch.float32, device=dev)
    dA = part[:, :Dm * N]
    dD = part[:, Dm * N:Dm * N + Dm] if want_dD else None
    dbias = part[:, pcols - Dm:] if want_db else None
    a = dm_scan_bwd_args()
    a.part_ss = pcols
    a.nseq, a.dim, a.seqlen, a.dstate = S, Dm, L, N
    a.ngroups = ngroups
    a.batch_per_dir = batch_per_dir
    a.io_dtype = dtype_code(u)
    a.bc_dtype = dtype_code(Bm)
    a.flags = flags
    a.ckpt_every = ckpt_every
    a.ckpt_dtype = _ckpt_dtype_code(ckpt)
    a.u, a.delta, a.z, a.dout = _ptr(u), _ptr(delta), _ptr(z), _ptr(dout)
    a.B, a.C, a.A, a.D, a.delta_bias = _ptr(Bm), _ptr(Cm), _ptr(A32), _ptr(D32), _ptr(b32)
    a.z_row_index, a.out_row_index = _ptr(z_row_index), _ptr(out_row_index)
    a.ckpt = _ptr(ckpt)
    a.du, a.ddelta, a.dz = _ptr(du), _ptr(ddelta), _ptr(dz)
    a.dBC_partial, a.dA_partial = _ptr(dBC), _ptr(dA)
    a.dD_partial, a.dbias_partial = _ptr(dD), _ptr(dbias)
    a.u_ss, a.u_sl, a.u_sd = u.stride()
    a.dt_ss, a.dt_sl, a.dt_sd = delta.stride()
    if z is not None:
        a.z_ss, a.z_sl, a.z_sd = z.stride()
        a.dz_ss, a.dz_sl, a.dz_sd = dz.stride()
    a.do_ss, a.do_sl, a.do_sd = dout.stride()
    a.B_ss, a.B_sl, a.B_sn = Bm.stride()
    a.C_ss, a.C_sl, a.C_sn = Cm.stride()
    a.B_sg = a.C_sg = N
    a.du_ss, a.du_sl, a.du_sd = du.stride()
    a.ddt_ss, a.ddt_sl, a.ddt_sd = ddelta.stride()
    nbytes = scan_bwd_algorithmic_bytes(S, Dm, L, N, u.element_size(), z is not None)
    design = nbytes + 2 * S * N * L * Bm.element_size() + S * L * (nw - 1) * 2 * N * 4 + 4 * S * Dm * (N + 2) \
        + (scan_nchunk(L, ckpt_every) - 1 + (L % ckpt_every == 0)) * S * ckpt.shape[2] * ckpt.shape[3] * 4   # slot 0 is read when L ends on a boundary
    _launch("dm_selective_scan_bwd", a, u, nbytes, design)
    if dbc_out is not None:                     # [S, L, 2N] view in the caller's buffer (the dB | dC columns of d x_dbl): summed + placed in one pass
        dBCs = sum_partials(dBC.view(S * L, nw, 2 * N), dbc_out)
    else:
        dBCs = sum_partials(dBC.view(S * L, nw, 2 * N), torch.empty((S, L, 2 * N), dtype=torch.float32, device=dev))   # deterministic
    dB, dC = dBCs[..., :N], dBCs[..., N:]
    psum = colsum(part)                         # [Dm*N | Dm | Dm]
    return (du, ddelta, dz, dB, dC, psum[:Dm * N].view(Dm, N), psum[Dm * N:Dm * N + Dm] if want_dD else None,
            psum[pcols - Dm:] if want_db else None)


def sum_partials(parts, out):
    """parts [M, nw, C] fp32 contiguous -> out[..., :C] = parts.sum(1), converted to out's dtype; out is any [.., C] tensor / view whose
    leading dims flatten to M rows of one stride (e.g. the dB | dC columns of the d x_dbl buffer).  Returns out."""
    M, nw, C = parts.shape
    o2 = out.reshape(M, C) if out.is_contiguous() else None
    if o2 is None:                              # a column block of a wider row-major buffer: rows must be evenly strided
        st = out.stride()
        sr = st[-2]
        ok = out.stride(-1) == 1 and all(st[i] == st[i + 1] * out.shape[i + 1] for i in range(out.dim() - 2))
        o2 = out.as_strided((M, C), (sr, 1)) if ok else None
    es = out.element_size()
    if (o2 is None or C % 4 or not parts.is_contiguous() or parts.dtype != torch.float32 or out.dtype not in _DT
            or o2.data_ptr() % (4 * es) or o2.stride(0) % 4):
        _pair_flush()
        out.copy_(parts.sum(dim=1).view(out.shape))
        return out
    a = dm_sum_partials_args()
    a.rows, a.nw, a.cols = M, nw, C
    a.out_dtype = _DT[out.dtype]
    setattr(a, "in", _ptr(parts))
    a.out = _ptr(o2)
    a.out_sr = o2.stride(0)
    _launch("dm_sum_partials", a, parts, M * C * (nw * 4 + es))
    return out


_COLSUM_SMALL = os.environ.get("DIFFMA_COLSUM_SMALL", "1") == "1"


def colsum(x, small=False):
    """x [R, C] fp32 contiguous -> [C] = x.sum(0) (dm_colsum_f32: ATen's outer-dimension reduction is 4x off HBM speed here)."""
    R, C = x.shape
    if C % 4 != 0 or not x.is_contiguous() or x.dtype != torch.float32 or (small and not _COLSUM_SMALL):
        _pair_flush()
        return x.sum(0)
    # The kernel gives every 256 columns ONE workgroup: a tall, narrow matrix (the conv backward's db partial rows at the Mamba-2 width:
    # 5 376 x 2 560 = 10 workgroups) is then read by a handful of CUs.  Fold rb row classes into the columns -- x[R][C] read as
    # [R / rb][rb C] is the same memory -- so that rb times as many workgroups stream it, and add the rb partial rows in a second,
    # tiny launch.  The summation order stays fixed by the shape.
    wgs = (C // 4 + 63) // 64
    if COLSUM_SPLIT and R >= 256 and wgs < 48:
        rb = 1
        while rb < 32 and R % (2 * rb) == 0 and wgs * rb < 192 and R // (2 * rb) >= 16:
            rb *= 2
        if rb > 1:
            return _colsum_launch(_colsum_launch(x.view(R // rb, rb * C)).view(rb, C))
    return _colsum_launch(x)


COLSUM_SPLIT = os.environ.get("DIFFMA_COLSUM_SPLIT", "1") == "1"        # 0: one workgroup per 256 columns whatever the shape (A/B runs)


def _colsum_launch(x):
    R, C = x.shape
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    a = dm_colsum_args()
    a.rows, a.cols = R, C
    setattr(a, "in", _ptr(x))
    a.out = _ptr(out)
    _launch("dm_colsum_f32", a, x, (R + 1) * C * 4)
    return out


def gather_conv1d_fwd(x, weight, bias, *, row_index=None, ndir=1, silu=True, out=None):
    """x: [B, L, Dm] token-major view (e.g. xz[..., :Dm]); weight [Dm, W]; -> [ndir*B, L, Dm]."""
    _require_gpu(x, weight, bias)
    Bsz, L, Dm = x.shape
    W = weight.shape[-1]
    weight = weight.reshape(Dm, W).contiguous()
    if weight.dtype != x.dtype and weight.dtype != torch.float32:
        weight = weight.float()
    if bias is not None:
        bias = bias.to(weight.dtype).contiguous()
    if out is None:
        out = torch.empty((ndir * Bsz, L, Dm), dtype=x.dtype, device=x.device)
    a = dm_conv_fwd_args()
    a.batch, a.dim, a.seqlen, a.width, a.ndir = Bsz, Dm, L, W, ndir
    a.io_dtype, a.w_dtype = dtype_code(x), dtype_code(weight)
    a.flags = DM_FLAG_SILU if silu else 0
    a.x, a.weight, a.bias, a.row_index, a.out = _ptr(x), _ptr(weight), _ptr(bias), _ptr(row_index), _ptr(out)
    a.x_sb, a.x_sl, a.x_sd = x.stride()
    a.o_ss, a.o_sl, a.o_sd = out.stride()
    _launch("dm_gather_conv1d_fwd", a, x, 2 * ndir * Bsz * L * Dm * x.element_size())
    return out


# Fused conv + x_proj (csrc/conv_xproj.hip): one workgroup walks one gathered sequence, so a launch needs two sequences per
# CU to fill the chip (measured, MI355X, D = 1024: 768 sequences 138 vs 207 us, 1536: 274 vs 398 us for the unfused pair,
# but 192: 58 vs 45 us); below that the unfused pair (chunk-parallel conv + library GEMM) is used.
XPROJ_FUSED_MIN_SEQS = 512


def conv_xproj_supported(x, wx, nseq, width=4):
    """True when dm_gather_conv1d_xproj_fwd serves this call (16-bit I/O, dim in {128..1024}, <= 64 projection rows, an
    instantiated conv width: once the fused path is chosen there is no fallback, so the predicate must know -- ADVICE r2)."""
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or nseq < XPROJ_FUSED_MIN_SEQS:
        return False
    if not _lib.load().dm_gather_conv1d_xproj_width_supported(int(width)):
        return False
    if x.stride(0) % 2 or x.stride(1) % 2 or x.storage_offset() % 2:
        return False
    return bool(_lib.load().dm_gather_conv1d_xproj_supported(int(x.shape[-1]), int(wx.shape[0]), dtype_code(x)))


def gather_conv1d_xproj_fwd(x, weight, bias, wx, *, row_index=None, ndir=1, silu=True):
    """x: [B, L, Dm] token-major view; weight [Dm, W]; wx [P, Dm] (x_proj.weight, same dtype as x).
    Returns (xc [ndir*B, L, Dm] = SiLU(conv(gathered x)), x_dbl [ndir*B*L, P] = xc @ wx^T) from ONE kernel."""
    _require_gpu(x, weight, bias, wx)
    Bsz, L, Dm = x.shape
    W = weight.shape[-1]
    P = wx.shape[0]
    weight = weight.reshape(Dm, W).contiguous()
    if weight.dtype != x.dtype and weight.dtype != torch.float32:
        weight = weight.float()
    if bias is not None:
        bias = bias.to(weight.dtype).contiguous()
    wx = wx.contiguous()
    assert wx.dtype == x.dtype and wx.shape[1] == Dm
    out = torch.empty((ndir * Bsz, L, Dm), dtype=x.dtype, device=x.device)
    xdbl = torch.empty((ndir * Bsz * L, P), dtype=x.dtype, device=x.device)
    a = dm_conv_xproj_fwd_args()
    a.batch, a.dim, a.seqlen, a.width, a.ndir = Bsz, Dm, L, W, ndir
    a.io_dtype, a.w_dtype = dtype_code(x), dtype_code(weight)
    a.flags = DM_FLAG_SILU if silu else 0
    a.nproj = P
    a.x, a.weight, a.bias, a.row_index = _ptr(x), _ptr(weight), _ptr(bias), _ptr(row_index)
    a.wx, a.out, a.xdbl = _ptr(wx), _ptr(out), _ptr(xdbl)
    a.x_sb, a.x_sl, a.x_sd = x.stride()
    a.o_ss, a.o_sl, a.o_sd = out.stride()
    a.xd_sr = P
    es = x.element_size()
    _launch("dm_gather_conv1d_xproj_fwd", a, x, 2 * ndir * Bsz * L * Dm * es + ndir * Bsz * L * P * es + P * Dm * es)
    return out, xdbl


# K4x (conv backward fused with d x~ = du + dx_dbl @ Wx; csrc/conv_xproj.hip): measured (MI355X, D = 1024, 1536 sequences)
# 532 us against 692 us for the in-place addmm + conv_bwd pair -- the product rides on the matrix pipe of a VALU-bound kernel
# and the 1.2 GB addmm pass disappears.  DIFFMA_FUSED_CONV_BWD=0 turns it off.
XPROJ_FUSED_BWD = os.environ.get("DIFFMA_FUSED_CONV_BWD", "1") == "1"


def conv_xproj_bwd_supported(x, wx, nseq, width=4, du=None, dxdbl=None):
    """True when dm_gather_conv1d_xproj_bwd serves this call (as the forward, plus a projection width that is a multiple of 8 and
    the alignment the C side checks on the gradient operands: du rows on 4-byte, dx_dbl rows on 16-byte boundaries)."""
    if not XPROJ_FUSED_BWD:
        return False
    if not _lib.load().dm_gather_conv1d_xproj_width_supported(int(width)):
        return False
    if du is not None and (du.data_ptr() % 4 or du.stride(0) % 2 or du.stride(1) % 2 or du.stride(2) != 1):
        return False
    if dxdbl is not None and (dxdbl.data_ptr() % 16 or dxdbl.stride(0) % 8 or dxdbl.stride(1) != 1):
        return False
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16) or nseq < XPROJ_FUSED_MIN_SEQS:
        return False
    if x.stride(0) % 2 or x.stride(1) % 2 or x.storage_offset() % 2:
        return False
    return bool(_lib.load().dm_gather_conv1d_xproj_bwd_supported(int(x.shape[-1]), int(wx.shape[0]), dtype_code(x)))


# DIFFMA_DX_MERGED=0: per-direction dx slabs + dm_token_merge again (A/B runs)
DX_MERGED = os.environ.get("DIFFMA_DX_MERGED", "1") == "1"


def gather_conv1d_xproj_bwd(x, weight, bias, du, dxdbl, wxt, *, row_index=None, ndir=1, silu=True, merged_out=None):
    """Conv backward whose incoming gradient is du + dxdbl @ wx, formed tile by tile inside the kernel (never materialised).
    x: [B, L, Dm] view; du: [ndir*B, L, Dm]; dxdbl: [ndir*B*L, P] (row stride any multiple of 8); wxt: [Dm, P] = x_proj.weight^T.
    Returns (dx_slabs [ndir*B, L, Dm] in token order, dweight [Dm, W] fp32, dbias [Dm] fp32).
    merged_out: a [B, L, Dm] view that receives the SUM of the directions' dx (DM_FLAG_DX_MERGED) -- returned in place of the slabs;
    no token_merge needed.  Two kernels behind it: sequences up to 256 rows take the slab form (a persistent workgroup per CU owns
    128 channels, the running sum lives in LDS, dx is written once); longer ones the whole-sample form (a workgroup walks the
    directions of a sample, the first stores, the others read-add-store)."""
    _require_gpu(x, weight, bias, du, dxdbl, wxt)
    Bsz, L, Dm = x.shape
    W = weight.shape[-1]
    P = wxt.shape[1]
    weight = weight.reshape(Dm, W).contiguous()
    if weight.dtype != x.dtype and weight.dtype != torch.float32:
        weight = weight.float()
    if bias is not None:
        bias = bias.to(weight.dtype).contiguous()
    wxt = wxt.contiguous()
    assert wxt.dtype == x.dtype and wxt.shape[0] == Dm and dxdbl.dtype == x.dtype and du.dtype == x.dtype
    assert dxdbl.stride(1) == 1 and du.stride(2) == 1
    S = ndir * Bsz
    dev = x.device
    merged = merged_out is not None
    if merged:
        assert merged_out.shape == (Bsz, L, Dm) and merged_out.dtype == x.dtype and merged_out.stride(2) == 1 and W == 4 and silu and row_index is not None
        dx, rows = merged_out, Bsz
    else:
        dx, rows = torch.empty((S, L, Dm), dtype=x.dtype, device=dev), S
    a = dm_conv_xproj_bwd_args()
    a.part_ss = Dm * (W + 1)
    a.batch, a.dim, a.seqlen, a.width, a.ndir = Bsz, Dm, L, W, ndir
    a.io_dtype, a.w_dtype = dtype_code(x), dtype_code(weight)
    a.flags = (DM_FLAG_SILU if silu else 0) | (DM_FLAG_DX_MERGED if merged else 0)
    a.nproj = P
    a.x, a.weight, a.bias, a.row_index = _ptr(x), _ptr(weight), _ptr(bias), _ptr(row_index)
    a.du, a.dxdbl, a.wxt = _ptr(du), _ptr(dxdbl), _ptr(wxt)
    a.dx = _ptr(dx)
    a.x_sb, a.x_sl, a.x_sd = x.stride()
    a.du_ss, a.du_sl, a.du_sd = du.stride()
    a.dx_ss, a.dx_sl, a.dx_sd = dx.stride()
    a.xd_sr = dxdbl.stride(0)
    # the slab form of the merged launch (running sum in LDS: no re-reads) sums dw | db per persistent workgroup stream: fewer partial rows
    slab_rows = int(_lib.load().dm_gather_conv1d_xproj_bwd_slab(ctypes.byref(a), None)) if merged else 0
    if slab_rows:
        rows = slab_rows
        a.flags |= DM_FLAG_PARTIAL_COMPACT
    part = torch.empty((rows, Dm * (W + 1)), dtype=torch.float32, device=dev)     # dw | db partial rows in one buffer: one column sum
    dw, db = part[:, :Dm * W], part[:, Dm * W:]
    a.dw_partial, a.db_partial = _ptr(dw), _ptr(db)
    es = x.element_size()
    # algorithmic bytes: x and du read per direction, dx written per direction (merged: written once -- the whole-sample form's
    # re-reads of the running sum are design traffic, mostly served by L2 / the Infinity Cache)
    nbytes = (2 * S + (Bsz if merged else S)) * L * Dm * es + S * L * P * es + P * Dm * es
    _launch("dm_gather_conv1d_xproj_bwd", a, x, nbytes, nbytes + (2 * (S - Bsz) * L * Dm * es if merged and not slab_rows else 0))
    psum = colsum(part)
    return dx, psum[:Dm * W].view(Dm, W), psum[Dm * W:]


def gather_conv1d_bwd(x, weight, bias, dout, *, row_index=None, ndir=1, silu=True):
    """Returns (dx_slabs [ndir*B, L, Dm] in token order, dweight [Dm, W] fp32, dbias [Dm] fp32)."""
    _require_gpu(x, weight, bias, dout)
    Bsz, L, Dm = x.shape
    W = weight.shape[-1]
    weight = weight.reshape(Dm, W).contiguous()
    if weight.dtype != x.dtype and weight.dtype != torch.float32:
        weight = weight.float()
    if bias is not None:
        bias = bias.to(weight.dtype).contiguous()
    lib = _lib.load()
    nchunk = lib.dm_conv_nchunk(L)
    dev = x.device
    dx = torch.empty((ndir * Bsz, L, Dm), dtype=x.dtype, device=dev)
    # dw | db partial rows in ONE buffer (part_ss): one column sum instead of two
    rows = ndir * Bsz * nchunk
    part = torch.empty((rows, Dm * (W + 1)), dtype=torch.float32, device=dev)
    dw, db = part[:, :Dm * W], part[:, Dm * W:]
    a = dm_conv_bwd_args()
    a.batch, a.dim, a.seqlen, a.width, a.ndir = Bsz, Dm, L, W, ndir
    a.io_dtype, a.w_dtype = dtype_code(x), dtype_code(weight)
    a.flags = DM_FLAG_SILU if silu else 0
    a.nchunk = nchunk
    a.x, a.weight, a.bias, a.dout = _ptr(x), _ptr(weight), _ptr(bias), _ptr(dout)
    a.row_index = _ptr(row_index)
    a.dx, a.dw_partial, a.db_partial = _ptr(dx), _ptr(dw), _ptr(db)
    a.x_sb, a.x_sl, a.x_sd = x.stride()
    a.do_ss, a.do_sl, a.do_sd = dout.stride()
    a.dx_ss, a.dx_sl, a.dx_sd = dx.stride()
    a.part_ss = Dm * (W + 1)
    _launch("dm_gather_conv1d_bwd", a, x, 3 * ndir * Bsz * L * Dm * x.element_size())
    # partial rows -> ONE column sum (ATen's reduction of these shapes takes ~21 us per call, dm_colsum_f32 ~5)
    psum = colsum(part, True)
    return dx, psum[:Dm * W].view(Dm, W), psum[Dm * W:]


def token_merge(slabs, *, row_index=None, out=None, out_dtype=None, gate=None, pre_out=None):
    """slabs: [K, B, L, Dm] (last stride 1) -> out [B, L, Dm] = sum_k slabs[k][:, idx_k, :].
    gate [B, L, Dm] view: out = sum * silu(gate) (the SiLU(z) gate of the K per-direction operators applied once per token);
    pre_out [B, L, Dm]: receives the ungated sum (saved for gate_bwd)."""
    _require_gpu(slabs, gate)
    K, Bsz, L, Dm = slabs.shape
    if out is None:
        out = torch.empty((Bsz, L, Dm), dtype=out_dtype or slabs.dtype, device=slabs.device)
    assert slabs.stride(3) == 1 and out.stride(2) == 1
    a = dm_merge_args()
    if gate is not None:
        assert gate.shape == (Bsz, L, Dm) and gate.stride(2) == 1 and gate.dtype == slabs.dtype and out.dtype == slabs.dtype
        a.gate = _ptr(gate)
        a.g_sb, a.g_sl = gate.stride()[:2]
        if pre_out is not None:
            assert pre_out.shape == (Bsz, L, Dm) and pre_out.stride(2) == 1 and pre_out.dtype == slabs.dtype
            a.pre = _ptr(pre_out)
            a.p_sb, a.p_sl = pre_out.stride()[:2]
    a.nin, a.batch, a.seqlen, a.dim = K, Bsz, L, Dm
    a.io_dtype, a.out_dtype = dtype_code(slabs), dtype_code(out)
    a.row_index = _ptr(row_index)
    a.out = _ptr(out)
    setattr(a, "in", _ptr(slabs))
    a.in_sk, a.in_sb, a.in_sl = slabs.stride()[:3]
    a.o_sb, a.o_sl = out.stride()[:2]
    extra = (0 if gate is None else 1) + (0 if pre_out is None else 1)
    _launch("dm_token_merge", a, slabs, ((K + extra) * slabs.element_size() + out.element_size()) * Bsz * L * Dm)
    return out


def repack(src, to_token_major, out=None):
    """The operator boundary's layout change as one HBM-bound pass (csrc/repack.hip).
    to_token_major=True : src (B, D, L) with stride(-1) == 1 (any batch / channel strides: CrossScan slices) -> [B, L, D] token-major;
    to_token_major=False: src [B, L, D] with stride(-1) == 1                                          -> (B, D, L) contiguous."""
    _require_gpu(src)
    assert src.dim() == 3 and src.stride(2) == 1
    a = dm_repack_args()
    if to_token_major:
        Bsz, Dm, L = src.shape
        if out is None:
            out = torch.empty((Bsz, L, Dm), dtype=src.dtype, device=src.device)
        cm, tm = src, out
    else:
        Bsz, L, Dm = src.shape
        if out is None:
            out = torch.empty((Bsz, Dm, L), dtype=src.dtype, device=src.device)
        cm, tm = out, src
    assert out.stride(2) == 1 and out.dtype == src.dtype and cm.shape == (Bsz, Dm, L) and tm.shape == (Bsz, L, Dm)
    a.batch, a.dim, a.seqlen = Bsz, Dm, L
    a.io_dtype = dtype_code(src)
    a.to_token_major = 1 if to_token_major else 0
    a.src, a.dst = _ptr(src), _ptr(out)
    a.cm_sb, a.cm_sd = cm.stride()[:2]
    a.tm_sb, a.tm_sl = tm.stride()[:2]
    _launch("dm_repack", a, src, 2 * Bsz * L * Dm * src.element_size())
    return out


def gate_bwd(dy, z, pre, *, dz_out=None):
    """Backward of y = pre * silu(z) (token_merge(..., gate=z)): returns (g = dy * silu(z), dz = dy * pre * silu'(z)).
    dy, pre: [B, L, Dm]; z: [B, L, Dm] view; dz_out: optional [B, L, Dm] view (e.g. the z half of d(xz))."""
    _require_gpu(dy, z, pre)
    Bsz, L, Dm = dy.shape
    assert z.shape == dy.shape and pre.shape == dy.shape and z.dtype == dy.dtype and pre.dtype == dy.dtype
    assert dy.stride(2) == 1 and z.stride(2) == 1 and pre.stride(2) == 1
    g = torch.empty((Bsz, L, Dm), dtype=dy.dtype, device=dy.device)
    dz = dz_out if dz_out is not None else torch.empty((Bsz, L, Dm), dtype=dy.dtype, device=dy.device)
    assert dz.stride(2) == 1 and dz.dtype == dy.dtype
    a = dm_gate_bwd_args()
    a.batch, a.seqlen, a.dim = Bsz, L, Dm
    a.io_dtype = dtype_code(dy)
    a.dy, a.z, a.pre, a.g, a.dz = _ptr(dy), _ptr(z), _ptr(pre), _ptr(g), _ptr(dz)
    a.dy_sb, a.dy_sl = dy.stride()[:2]
    a.z_sb, a.z_sl = z.stride()[:2]
    a.p_sb, a.p_sl = pre.stride()[:2]
    a.g_sb, a.g_sl = g.stride()[:2]
    a.dz_sb, a.dz_sl = dz.stride()[:2]
    _launch("dm_gate_bwd", a, dy, 5 * Bsz * L * Dm * dy.element_size())
    return g, dz


# dt_proj with softplus in the epilogue (csrc/dtproj.hip); DIFFMA_DTPROJ_FUSED=0 returns the mixer to F.linear + softplus in the scans
DTPROJ_FUSED = os.environ.get("DIFFMA_DTPROJ_FUSED", "1") == "1"


def dtproj_softplus_supported(xdbl, w):
    if not (DTPROJ_FUSED and xdbl.is_cuda and xdbl.dtype in (torch.bfloat16, torch.float16) and w.dtype == xdbl.dtype):
        return False
    if xdbl.stride(-1) != 1 or xdbl.stride(0) % 8 or xdbl.data_ptr() % 16:
        return False
    return bool(_lib.load().dm_dtproj_softplus_supported(int(w.shape[0]), int(w.shape[1]), dtype_code(xdbl)))


def dtproj_softplus_fwd(xdbl, w, bias):
    """xdbl: [M, >= R] (row stride a multiple of 8, 16-bit), w: [Dm, R] (dt_proj.weight, same dtype), bias [Dm] fp32 or None.
    Returns delta [M, Dm] = softplus(xdbl[:, :R] @ w^T + bias) in the I/O dtype."""
    _require_gpu(xdbl, w, bias)
    M = xdbl.shape[0]
    Dm, R = w.shape
    w = w.contiguous()
    bias = _f32c(bias)
    delta = torch.empty((M, Dm), dtype=xdbl.dtype, device=xdbl.device)
    a = dm_dtproj_args()
    a.rows, a.dim, a.rank = M, Dm, R
    a.io_dtype = dtype_code(xdbl)
    a.xdbl, a.w, a.bias, a.delta = _ptr(xdbl), _ptr(w), _ptr(bias), _ptr(delta)
    a.xd_sr = xdbl.stride(0)
    es = xdbl.element_size()
    _launch("dm_dtproj_softplus_fwd", a, xdbl, M * (Dm + R) * es + Dm * R * es + 4 * Dm)
    return delta


DTPROJ_BWD_BLOCKS = 256              # workgroups (= partial dW images) of dm_dtproj_bwd: one per CU


def dtproj_bwd_supported(ddelta, xdbl, w, dxdbl):
    """ddelta [M, Dm] contiguous, xdbl / dxdbl [M, >= R] row-major views (16-bit), w [Dm, R]."""
    if not (DTPROJ_FUSED and ddelta.is_cuda and ddelta.dtype in (torch.bfloat16, torch.float16) and w.dtype == ddelta.dtype
            and xdbl.dtype == ddelta.dtype and dxdbl.dtype == ddelta.dtype):
        return False
    if ddelta.shape[0] < 1 or not ddelta.is_contiguous() or ddelta.data_ptr() % 16:
        return False
    if xdbl.stride(-1) != 1 or xdbl.stride(0) % 8 or xdbl.data_ptr() % 16 or dxdbl.stride(-1) != 1 or dxdbl.stride(0) % 4 or dxdbl.data_ptr() % 8:
        return False
    return bool(_lib.load().dm_dtproj_bwd_supported(int(w.shape[0]), int(w.shape[1]), dtype_code(ddelta)))


def dtproj_bwd(ddelta, xdbl, w, dxdbl):
    """One read of ddelta [M, Dm]: dxdbl[:, :R] = ddelta @ w (written in place into the given rows) and returns
    dW [Dm, R] fp32 = ddelta^T @ xdbl[:, :R]."""
    _require_gpu(ddelta, xdbl, w, dxdbl)
    M, Dm = ddelta.shape
    R = w.shape[1]
    w = w.contiguous()
    nblk = max(1, min(DTPROJ_BWD_BLOCKS, (M + 31) // 32))
    part = torch.empty((nblk, Dm * R), dtype=torch.float32, device=ddelta.device)
    a = dm_dtproj_bwd_args()
    a.rows, a.dim, a.rank, a.io_dtype, a.nblk = M, Dm, R, dtype_code(ddelta), nblk
    a.ddelta, a.xdbl, a.w, a.dxdbl, a.part = _ptr(ddelta), _ptr(xdbl), _ptr(w), _ptr(dxdbl), _ptr(part)
    a.xd_sr, a.dxd_sr = xdbl.stride(0), dxdbl.stride(0)
    es = ddelta.element_size()
    _launch("dm_dtproj_bwd", a, ddelta, M * (Dm + 2 * R) * es + Dm * R * (es + 4))
    return colsum(part).view(Dm, R)


# ------------------------------------------------------------------------------------------------
# Dense products of the projections in the small-launch regime (csrc/gemm.hip)
# ------------------------------------------------------------------------------------------------
def _gemm_dims(a, b, a_kmajor, b_kmajor):
    P, Kc = (a.shape[0], a.shape[1]) if a_kmajor else (a.shape[1], a.shape[0])
    Q, Kb = (b.shape[0], b.shape[1]) if b_kmajor else (b.shape[1], b.shape[0])
    if Kb != Kc:
        raise ValueError(f"contraction sizes differ: {Kc} vs {Kb}")
    return P, Q, Kc


def gemm_supported(a, b, a_kmajor, b_kmajor, out_dtype=None):
    """a, b: 2-D stored matrices (column stride 1).  See include/diffma_hip.h dm_gemm_args."""
    if not (a.is_cuda and a.dtype in (torch.bfloat16, torch.float16) and b.dtype == a.dtype and a.dim() == 2 and b.dim() == 2):
        return False
    if a.stride(1) != 1 or b.stride(1) != 1 or a.stride(0) % 8 or b.stride(0) % 8 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return False
    P, Q, Kc = _gemm_dims(a, b, a_kmajor, b_kmajor)
    cd = _DT[out_dtype or a.dtype]
    return bool(_lib.load().dm_gemm_supported(P, Q, Kc, int(a_kmajor), int(b_kmajor), dtype_code(a), cd))


def gemm(a, b, a_kmajor=True, b_kmajor=True, out=None, out_dtype=None, accumulate=False):
    """C[P, Q] = opA(a) @ opB(b) on the matrix pipe, fp32 accumulation (csrc/gemm.hip).  a_kmajor: a is [P, Kc], else [Kc, P];
    b_kmajor: b is [Q, Kc], else [Kc, Q].  forward: gemm(x, W); dgrad: gemm(dy, W, True, False); wgrad: gemm(dy, x, False, False,
    out_dtype=torch.float32)."""
    _require_gpu(a, b)
    P, Q, Kc = _gemm_dims(a, b, a_kmajor, b_kmajor)
    if out is None:
        out = torch.empty((P, Q), dtype=out_dtype or a.dtype, device=a.device)
    g = dm_gemm_args()
    g.P, g.Q, g.Kc = P, Q, Kc
    g.ab_dtype, g.c_dtype = dtype_code(a), dtype_code(out)
    g.a_kmajor, g.b_kmajor = int(a_kmajor), int(b_kmajor)
    g.accumulate = int(bool(accumulate))
    g.a, g.b, g.c = _ptr(a), _ptr(b), _ptr(out)
    g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
    _launch("dm_gemm", g, a, (P * Kc + Q * Kc) * a.element_size() + P * Q * out.element_size(), flops=2 * P * Q * Kc)
    return out


# ------------------------------------------------------------------------------------------------
# Dense products of the projections at LARGE batch (csrc/gemm_large.hip, K12): persistent 256 x 256 tiles, both operands k-major
# ------------------------------------------------------------------------------------------------
GEMM_LARGE = os.environ.get("DIFFMA_GEMM_LARGE", "1") == "1"      # 0: the library GEMMs again (A/B runs)


def gemm_large_supported(a, b, out=None):
    """a [P, Kc], b [Q, Kc] stored matrices (column stride 1), 16-bit, for C = a @ b^T in the operand dtype."""
    if not (GEMM_LARGE and a.is_cuda and a.dtype in (torch.bfloat16, torch.float16) and b.dtype == a.dtype and a.dim() == 2 and b.dim() == 2):
        return False
    if a.stride(1) != 1 or b.stride(1) != 1 or a.stride(0) % 8 or b.stride(0) % 8 or a.data_ptr() % 16 or b.data_ptr() % 16:
        return False
    if a.shape[1] != b.shape[1]:
        return False
    if out is not None and (out.dtype != a.dtype or out.stride(1) != 1 or out.stride(0) % 8 or out.data_ptr() % 16):
        return False
    big = max((a.shape[0] + 256) * a.stride(0), b.shape[0] * b.stride(0), (a.shape[0] + 256) * (out.stride(0) if out is not None else b.shape[0]))
    if big * 2 >= 0x7ffffff0:
        return False
    code = dtype_code(a)
    return bool(_lib.load().dm_gemm_large_supported(a.shape[0], b.shape[0], a.shape[1], 1, 1, code, code))


def gemm_large(a, b, out=None):
    """C[P, Q] = a @ b^T (a [P, Kc], b [Q, Kc]) by the persistent large-batch kernel; the caller checked gemm_large_supported."""
    _require_gpu(a, b)
    P, Kc = a.shape
    Q = b.shape[0]
    if out is None:
        out = torch.empty((P, Q), dtype=a.dtype, device=a.device)
    g = dm_gemm_args()
    g.P, g.Q, g.Kc = P, Q, Kc
    g.ab_dtype, g.c_dtype = dtype_code(a), dtype_code(out)
    g.a_kmajor, g.b_kmajor = 1, 1
    g.accumulate = 0
    g.a, g.b, g.c = _ptr(a), _ptr(b), _ptr(out)
    g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
    _launch("dm_gemm_large", g, a, (P * Kc + Q * Kc) * a.element_size() + P * Q * out.element_size(), flops=2 * P * Q * Kc)
    return out


LN_ROWS_PER_BLOCK = 28   # DM_LN_ROWS_PER_BLOCK
LN_SMALL_ROWS = int(os.environ.get("DIFFMA_LN_SMALL_ROWS", "4"))     # rows per partial-sum group of the backward when a launch has few rows; 0: always 28


def _ln_rows_per_block(Bsz, L):
    """Rows per workgroup (= per partial row) of dm_ln_mod_bwd / dm_blend_bwd.  A workgroup's 4 waves walk their rows one after the
    other; with 28 rows a launch of few samples (the reference trains at ONE per GPU: 7 workgroups) is a chain of seven memory
    latencies on 7 CUs.  Below one workgroup per CU: 4 rows (one per wave), 49 workgroups per sample at L = 196."""
    if LN_SMALL_ROWS > 0 and Bsz * ((L + LN_ROWS_PER_BLOCK - 1) // LN_ROWS_PER_BLOCK) < 256:
        return LN_SMALL_ROWS
    return LN_ROWS_PER_BLOCK


def _ln_args(x, x2, gamma, beta, shift, scale, mask, eps, y_dtype):
    Bsz, L, C1 = x.shape
    C2 = x2.shape[-1] if x2 is not None else 0
    a = dm_ln_mod_args()
    a.batch, a.rows_per_batch, a.C1, a.C2 = Bsz, L, C1, C2
    a.x_dtype, a.y_dtype = dtype_code(x), _DT[y_dtype]
    mod = scale if scale is not None else mask
    a.mod_dtype = dtype_code(mod) if mod is not None else DM_F32
    a.eps = eps
    a.x, a.x2, a.gamma, a.beta = _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta)
    a.shift, a.scale, a.mask = _ptr(shift), _ptr(scale), _ptr(mask)
    a.x_sr = x.stride(1)
    a.x2_sr = x2.stride(1) if x2 is not None else 0
    a.y_sr = C1 + C2
    a.mod_sb = scale.stride(0) if scale is not None else 0
    return a, Bsz, L, C1, C2


def ln_mod_fwd(x, x2, gamma, beta, shift, scale, mask, eps, y_dtype):
    """LayerNorm([x|x2])*gamma+beta -> optional modulate(shift, scale per batch) -> (y1, y1*mask or None, stats)."""
    _require_gpu(x, x2, gamma, beta, shift, scale, mask)
    a, Bsz, L, C1, C2 = _ln_args(x, x2, gamma, beta, shift, scale, mask, eps, y_dtype)
    C = C1 + C2
    if mask is not None:        # the two halves of ONE buffer: the paired mixers' in_proj reads them as a batch of 2 without a copy
        y12 = torch.empty((2, Bsz, L, C), dtype=y_dtype, device=x.device)
        y1, y2 = y12[0], y12[1]
    else:
        y1, y2 = torch.empty((Bsz, L, C), dtype=y_dtype, device=x.device), None
    stats = torch.empty((Bsz * L, 2), dtype=torch.float32, device=x.device)
    a.y1, a.y2, a.stats = _ptr(y1), _ptr(y2), _ptr(stats)
    esz = x.element_size()
    _launch("dm_ln_mod_fwd", a, x, Bsz * L * C * (esz + y1.element_size() * (2 if mask is not None else 1)))
    return y1, y2, stats


def ln_mod_bwd(x, x2, gamma, beta, shift, scale, mask, eps, stats, dy1, dy2, dx=None, dx2=None, accumulate=False, dx_add=None, mod_dtype=None):
    """Returns (dx, dx2, dshift, dscale, dgamma, dbeta); dx/dx2 may be given (accumulate=True adds into them); dx_add [B, L, C1]
    (x's dtype, read-only): dx = computed + dx_add -- the gradient of x's other consumer, which may be shared and stays intact.
    mod_dtype: dtype of the returned dshift / dscale (default fp32): the partial rows are summed INTO that dtype (fp32 accumulation
    inside the reduction) instead of being summed and cast in two launches each -- the small-batch step is made of launches."""
    _require_gpu(x, dy1)
    a, Bsz, L, C1, C2 = _ln_args(x, x2, gamma, beta, shift, scale, mask, eps, dy1.dtype)
    C = C1 + C2
    if dx is None:
        dx = torch.empty((Bsz, L, C1), dtype=x.dtype, device=x.device)
    if x2 is not None and dx2 is None:
        dx2 = torch.empty((Bsz, L, C2), dtype=x2.dtype, device=x.device)
    a.rows_per_block = rpb = _ln_rows_per_block(Bsz, L)
    bpb = (L + rpb - 1) // rpb
    part = torch.empty((Bsz, bpb, 4, C), dtype=torch.float32, device=x.device)
    a.stats, a.dy1, a.dy2, a.dx, a.dx2, a.part = _ptr(stats), _ptr(dy1), _ptr(dy2), _ptr(dx), _ptr(dx2), _ptr(part)
    a.dx_sr = dx.stride(1)
    a.dx2_sr = dx2.stride(1) if dx2 is not None else 0
    a.accumulate = 1 if accumulate else 0
    if dx_add is not None:
        assert x2 is None and dx_add.shape == x.shape and dx_add.dtype == x.dtype and dx_add.stride(2) == 1 and dx_add.stride(0) == L * dx_add.stride(1)
        a.dx_add, a.dxa_sr = _ptr(dx_add), dx_add.stride(1)
    assert dy1.is_contiguous() and (dy2 is None or dy2.is_contiguous())
    # algorithmic bytes: x read, dx written, the incoming gradient(s) read -- and the gradient dx is added to (dx_add / accumulate),
    # which any implementation of "dx = f(x, dy) + other" has to read as well
    has_old = dx_add is not None or bool(accumulate)
    _launch("dm_ln_mod_bwd", a, x, Bsz * L * C * ((3 if has_old else 2) * x.element_size() + dy1.element_size() * (2 if dy2 is not None else 1)))
    if scale is None and shift is None:       # no modulation (the LayerNorm of the fusion MLP): only d gamma / d beta are wanted -- one
        pg = colsum(part.view(Bsz * bpb, 4 * C), True).view(4, C)      # column sum over all partial rows instead of two reductions
        return dx, dx2, None, None, pg[2], pg[3]
    if mod_dtype is not None and mod_dtype != torch.float32:
        ds = part[:, :, :2].sum(1, dtype=mod_dtype)                    # [B, 2, C] in the consumer's dtype: one launch
        pg = part[:, :, 2:].sum((0, 1))                                # [2, C] fp32
        return dx, dx2, ds[:, 0], ds[:, 1], pg[0], pg[1]
    pb = part.sum(1)                          # [B, 4, C]
    dshift, dscale = pb[:, 0], pb[:, 1]
    pg = pb.sum(0)
    return dx, dx2, dshift, dscale, pg[2], pg[3]


def _blend_args(x_like, xs, ws, a_row, gate):
    Bsz, L, C = xs.shape
    a = dm_blend_args()
    a.batch, a.rows_per_batch, a.C = Bsz, L, C
    a.x_dtype, a.s_dtype, a.g_dtype = dtype_code(x_like), dtype_code(xs), dtype_code(gate)
    a.xs, a.ws, a.a, a.gate = _ptr(xs), _ptr(ws), _ptr(a_row), _ptr(gate)
    a.gate_sb = gate.stride(0)
    return a, Bsz, L, C


def blend_fwd(x, xs, ws, a_row, gate):
    """x + gate[b] * (a*xs + (1-a)*ws);  x [B,L,C] (contiguous), xs/ws [B,L,C], a_row [B,L,1], gate [B,C] view."""
    _require_gpu(x, xs, ws, a_row, gate)
    a, Bsz, L, C = _blend_args(x, xs, ws, a_row, gate)
    out = torch.empty_like(x)
    a.x, a.out = _ptr(x), _ptr(out)
    _launch("dm_blend_fwd", a, x, Bsz * L * C * (2 * x.element_size() + 2 * xs.element_size()))
    return out


def blend_bwd(g, xs, ws, a_row, gate):
    """Returns (dxs, dws, da [B,L,1], dgate [B,C] in the dtype of gate: the partial rows are summed into it, fp32 accumulation)."""
    _require_gpu(g, xs, ws, a_row, gate)
    a, Bsz, L, C = _blend_args(g, xs, ws, a_row, gate)
    if xs.shape == ws.shape and xs.dtype == ws.dtype and xs.is_contiguous() and ws.is_contiguous():
        d2 = torch.empty((2,) + tuple(xs.shape), dtype=xs.dtype, device=xs.device)      # halves of one buffer: the paired mixers'
        dxs, dws = d2[0], d2[1]                                                          # out_proj backward reads them as a batch of 2
    else:
        dxs, dws = torch.empty_like(xs), torch.empty_like(ws)
    da = torch.empty_like(a_row)
    a.rows_per_block = rpb = _ln_rows_per_block(Bsz, L)
    bpb = (L + rpb - 1) // rpb
    part = torch.empty((Bsz, bpb, C), dtype=torch.float32, device=g.device)
    a.g, a.dxs, a.dws, a.da, a.dgate_part = _ptr(g), _ptr(dxs), _ptr(dws), _ptr(da), _ptr(part)
    _launch("dm_blend_bwd", a, g, Bsz * L * C * (g.element_size() + 4 * xs.element_size()))
    return dxs, dws, da, part.sum(1, dtype=gate.dtype)


# ------------------------------------------------------------------------------------------------
# Tail of the block's fusion MLP: bias + SiLU + Linear(C, 1) + Sigmoid in one pass (csrc/gate_head.hip)
# ------------------------------------------------------------------------------------------------
GATE_HEAD_BWD_BLOCKS = 1024          # partial rows of the backward (= its grid: 4 workgroups per CU)


def gate_head_supported(h, C=None):
    """16-byte rows: C a multiple of 4 (fp32) / 8 (16-bit) and at most 1024 / 2048 values."""
    C = h.shape[-1] if C is None else C
    vec = 16 // h.element_size()
    return h.is_cuda and h.dtype in _DT and C % vec == 0 and C <= 64 * vec * 4 and h.stride(-1) == 1


def _gate_head_args(h2, b1, w2, b2, a_row):
    a = dm_gate_head_args()
    a.rows, a.C, a.io_dtype = h2.shape[0], h2.shape[1], dtype_code(h2)
    a.h, a.h_sr = _ptr(h2), h2.stride(0)
    a.b1, a.w2, a.b2, a.a = _ptr(b1), _ptr(w2), _ptr(b2), _ptr(a_row)
    return a


def gate_head_fwd(h, b1, w2, b2):
    """h [..., C] (the first Linear's output WITHOUT its bias); b1 [C] / None, w2 [C], b2 [1] / None in fp32 -> a [..., 1] (dtype of h)
    = sigmoid(silu(h + b1) @ w2 + b2)."""
    _require_gpu(h, b1, w2, b2)
    C = h.shape[-1]
    h2 = h.reshape(-1, C)
    a_row = torch.empty(h.shape[:-1] + (1,), dtype=h.dtype, device=h.device)
    a = _gate_head_args(h2, b1, w2, b2, a_row)
    _launch("dm_gate_head_fwd", a, h, h2.numel() * h.element_size() + a_row.numel() * h.element_size())
    return a_row


def gate_head_bwd(da, a_row, h, b1, w2):
    """-> (dh like h, db1 [C] fp32, dw2 [C] fp32, db2 [1] fp32)."""
    _require_gpu(da, a_row, h, b1, w2)
    C = h.shape[-1]
    h2 = h.reshape(-1, C)
    da = da.reshape(-1)
    if da.dtype != h.dtype or not da.is_contiguous():
        da = da.to(h.dtype).contiguous()
    dh = torch.empty(h2.shape, dtype=h.dtype, device=h.device)
    nblk = max(1, min(GATE_HEAD_BWD_BLOCKS, (h2.shape[0] + 3) // 4))
    part = torch.empty((nblk, 2 * C + 4), dtype=torch.float32, device=h.device)
    a = _gate_head_args(h2, b1, w2, None, a_row)
    a.da, a.dh, a.dh_sr, a.part, a.nblk = _ptr(da), _ptr(dh), dh.stride(0), _ptr(part), nblk
    _launch("dm_gate_head_bwd", a, h, 2 * h2.numel() * h.element_size() + 2 * da.numel() * h.element_size())
    sums = colsum(part)
    return dh.view(h.shape), sums[:C], sums[C:2 * C], sums[2 * C:2 * C + 1]


# ------------------------------------------------------------------------------------------------
# Mamba-2 epilogue: gated RMSNorm of every direction slab + 3-way merge (csrc/rmsnorm.hip)
# ------------------------------------------------------------------------------------------------
def _rms_args(y, weight, eps):
    K, Bsz, L, C = y.shape
    assert y.stride(3) == 1 and y.stride(1) == L * y.stride(2), "slabs must be [K, B, L, C] with contiguous rows"
    a = dm_rmsnorm_merge_args()
    a.nslab, a.C, a.rows = K, C, Bsz * L
    a.io_dtype = dtype_code(y)
    a.eps = float(eps)
    a.y, a.weight = _ptr(y), _ptr(weight)
    a.y_ss, a.y_sr = y.stride(0), y.stride(2)
    return a, K, Bsz, L, C


def rmsnorm_merge_fwd(y, weight, eps):
    """y [K, B, L, C] (token order, gated) -> (out [B, L, C] = w * sum_k rmsnorm(y_k), rstd [K, B*L] fp32)."""
    weight = _f32c(weight)
    _require_gpu(y, weight)
    a, K, Bsz, L, C = _rms_args(y, weight, eps)
    out = torch.empty((Bsz, L, C), dtype=y.dtype, device=y.device)
    rstd = torch.empty((K, Bsz * L), dtype=torch.float32, device=y.device)
    a.out, a.rstd, a.out_sr = _ptr(out), _ptr(rstd), C
    _launch("dm_rmsnorm_merge_fwd", a, y, (K + 1) * Bsz * L * C * y.element_size())
    return out, rstd


def rmsnorm_merge_bwd(y, weight, eps, rstd, dout):
    """Returns (dy [K, B, L, C], dweight [C] fp32)."""
    weight = _f32c(weight)
    _require_gpu(y, weight, dout)
    a, K, Bsz, L, C = _rms_args(y, weight, eps)
    dout = dout.contiguous()
    dy = torch.empty((K, Bsz, L, C), dtype=y.dtype, device=y.device)
    rpb = _lib.load().dm_rmsnorm_merge_rows_per_block()
    nblk = (Bsz * L + rpb - 1) // rpb
    part = torch.empty((nblk, C), dtype=torch.float32, device=y.device)
    a.rstd, a.dout, a.dy, a.dw_part = _ptr(rstd), _ptr(dout), _ptr(dy), _ptr(part)
    a.dout_sr, a.dy_ss, a.dy_sr = C, dy.stride(0), C
    _launch("dm_rmsnorm_merge_bwd", a, y, (2 * K + 1) * Bsz * L * C * y.element_size())
    return dy, colsum(part, True)


# ------------------------------------------------------------------------------------------------
# Sampler step after the denoiser call (csrc/diffusion_step.hip)
# ------------------------------------------------------------------------------------------------
def diffusion_step(model_out, x, t, noise, tables, rows, *, ddim=False, eta=0.0, clip=False):
    """model_out [B, 2C, ...] (eps | variance logits), x / noise [B, C, ...] fp32, t [B] int64, tables fp32 [K, T];
    rows: the 8 row numbers (posterior log variance, log betas, sqrt(1/abar), sqrt(1/abar - 1), coef1, coef2, abar, abar_prev).
    Returns (sample, pred_xstart), both fp32 like x: one kernel instead of ~25 elementwise launches."""
    _require_gpu(model_out, x, t, noise, tables)
    B, C = x.shape[:2]
    hw = x[0, 0].numel()
    model_out = model_out.contiguous()
    x = x.contiguous()
    noise = noise.contiguous() if noise is not None else None
    assert model_out.shape[1] == 2 * C and x.dtype == torch.float32 and t.dtype == torch.int64 and tables.dtype == torch.float32
    sample, x0 = torch.empty_like(x), torch.empty_like(x)
    a = dm_diffusion_step_args()
    a.batch, a.channels, a.hw, a.T = B, C, hw, tables.shape[1]
    a.mode, a.clip, a.out_dtype = (1 if ddim else 0), (1 if clip else 0), dtype_code(model_out)
    a.eta = float(eta)
    (a.row_post_logvar, a.row_log_betas, a.row_sqrt_recip_ac, a.row_sqrt_recipm1_ac, a.row_coef1, a.row_coef2, a.row_ac,
     a.row_ac_prev) = rows
    a.model_out, a.x, a.noise, a.t, a.tables = _ptr(model_out), _ptr(x), _ptr(noise), _ptr(t), _ptr(tables)
    a.sample, a.pred_xstart = _ptr(sample), _ptr(x0)
    _launch("dm_diffusion_step", a, x, B * C * hw * (2 * model_out.element_size() + 4 * 4))
    return sample, x0


def _loss_args(x, t, tables, rows, out_dtype_code):
    B, C = x.shape[:2]
    a = dm_training_loss_args()
    a.batch, a.channels, a.hw, a.T = B, C, x[0, 0].numel(), tables.shape[1]
    a.out_dtype = out_dtype_code
    (a.row_sqrt_ac, a.row_sqrt_1mac, a.row_post_logvar, a.row_log_betas, a.row_sqrt_recip_ac, a.row_sqrt_recipm1_ac, a.row_coef1,
     a.row_coef2) = rows
    a.t, a.tables = _ptr(t), _ptr(tables)
    return a


def q_sample(x_start, noise, t, tables, rows):
    """x_t = sqrt(abar_t) x_start + sqrt(1 - abar_t) noise in one launch (csrc/diffusion_loss.hip); fp32 (B, C, ...), t int64 [B]."""
    _require_gpu(x_start, noise, t, tables)
    assert x_start.dtype == torch.float32 and noise.dtype == torch.float32 and t.dtype == torch.int64 and tables.dtype == torch.float32
    x_start, noise = x_start.contiguous(), noise.contiguous()
    out = torch.empty_like(x_start)
    a = _loss_args(x_start, t, tables, rows, DM_F32)
    a.x_start, a.noise, a.x_t_out = _ptr(x_start), _ptr(noise), _ptr(out)
    _launch("dm_q_sample", a, x_start, 3 * x_start.numel() * 4)
    return out


def training_loss_fwd(model_out, x_start, x_t, noise, t, tables, rows):
    """(mse [B], vb [B], loss [B], grad fp32 [B, 2C, ...]): the per-sample terms of GaussianDiffusion.training_losses and their gradient
    with respect to the model output, one launch (workgroup = sample)."""
    _require_gpu(model_out, x_start, x_t, noise, t, tables)
    B, C = x_start.shape[:2]
    assert model_out.shape[0] == B and model_out.shape[1] == 2 * C and model_out.is_contiguous()
    assert all(v.dtype == torch.float32 and v.is_contiguous() for v in (x_start, x_t, noise)) and t.dtype == torch.int64
    dev = x_start.device
    terms = torch.empty((3, B), dtype=torch.float32, device=dev)
    grad = torch.empty(model_out.shape, dtype=torch.float32, device=dev)
    a = _loss_args(x_start, t, tables, rows, dtype_code(model_out))
    a.model_out, a.x_start, a.x_t, a.noise = _ptr(model_out), _ptr(x_start), _ptr(x_t), _ptr(noise)
    a.mse, a.vb, a.loss, a.grad = _ptr(terms[0]), _ptr(terms[1]), _ptr(terms[2]), _ptr(grad)
    _launch("dm_training_loss", a, x_start, model_out.numel() * (model_out.element_size() + 4) + 3 * x_start.numel() * 4)
    return terms[0], terms[1], terms[2], grad


def training_loss_bwd(grad, g_eps, g_v, out_dtype):
    """grad_out (dtype of the model output) = per-sample scalars times the stored gradient: (g_eps[b] * grad[b, :C] | g_v[b] * grad[b, C:])."""
    _require_gpu(grad, g_eps, g_v)
    B, C2 = grad.shape[:2]
    out = torch.empty(grad.shape, dtype=out_dtype, device=grad.device)
    a = dm_training_loss_args()
    a.batch, a.channels, a.hw, a.T = B, C2 // 2, grad[0, 0].numel(), 1
    a.out_dtype = _DT[out_dtype]
    g_eps, g_v = g_eps.float().contiguous(), g_v.float().contiguous()
    a.grad, a.g_eps, a.g_v, a.grad_out = _ptr(grad), _ptr(g_eps), _ptr(g_v), _ptr(out)
    _launch("dm_training_loss_bwd", a, grad, grad.numel() * (4 + out.element_size()))
    return out


# ------------------------------------------------------------------------------------------------
# Mamba-2 SSD core on the matrix pipe (csrc/ssd.hip, csrc/ssd_bwd.hip): --use-mamba2 with 16-bit activations
# ------------------------------------------------------------------------------------------------
# On by default ("keep it only if it beats the A-shared scan" -- it does, DiffMa-XL/2
# mixer shape, 16 heads x 64, L 196: 26 vs 68 us at nseq 24, 101 vs 183 us at nseq 192, 415 vs 513 us at nseq 768, tools/bench_ssd.py).
# One wave per (sequence, head, 32-column half); the decay is factorised per tile pair so only the diagonal tiles take
# element-wise exps; X / z / out tiles move as 16-byte row pieces through LDS.  DIFFMA_SSD_MFMA=0 returns the no-grad Mamba-2
# mixer to the A-shared scan.  fp32 I/O stays on the scan; the backward twin is ssd_bwd below.
SSD_MFMA = os.environ.get("DIFFMA_SSD_MFMA", "1") == "1"


def ssd_fwd_supported(x, L, headdim, dstate, views=()):
    """views: the x / B / C / z views the launch would get -- their rows must start on 16-byte boundaries (tiles move as 16-byte pieces)."""
    if not (SSD_MFMA and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16)):
        return False
    for v in views:
        if v is not None and (v.data_ptr() % 16 or v.stride(-1) != 1 or any(st % 8 for st in v.stride()[:-1])):
            return False
    return bool(_lib.load().dm_ssd_fwd_supported(int(L), int(headdim), int(dstate), dtype_code(x)))


def ssd_fwd(x, Bm, Cm, dt_tok, z, A_h, D_h, dt_bias_h, *, z_row_index=None, out_row_index=None, batch_per_dir=0, out=None):
    """x: [S, L, H*64] view, Bm / Cm: [S, L, 16] views (after the conv, per gathered sequence); dt_tok: [B, L, H] and z: [B, L, H*64]
    in token order; A_h, D_h, dt_bias_h: [H] fp32.  Returns the gated output [S, L, H*64] with step l at row out_row_index[dir][l]."""
    _require_gpu(x, Bm, Cm, dt_tok, z)
    S, L, Din = x.shape
    H = A_h.shape[0]
    assert Din == H * 64 and Bm.shape[-1] == 16 and x.stride(2) == 1 and Bm.stride(2) == 1 and Cm.stride(2) == 1 and dt_tok.stride(2) == 1
    if out is None:
        out = torch.empty((S, L, Din), dtype=x.dtype, device=x.device)
    A32, D32, b32 = _f32c(A_h), _f32c(D_h), _f32c(dt_bias_h)
    a = dm_ssd_fwd_args()
    a.nseq, a.batch_per_dir, a.seqlen, a.nheads, a.headdim, a.dstate = S, batch_per_dir, L, H, 64, 16
    a.io_dtype, a.flags = dtype_code(x), 0
    a.x, a.B, a.C, a.dt, a.z = _ptr(x), _ptr(Bm), _ptr(Cm), _ptr(dt_tok), _ptr(z)
    a.A, a.D, a.dt_bias = _ptr(A32), _ptr(D32), _ptr(b32)
    a.z_row_index, a.out_row_index = _ptr(z_row_index), _ptr(out_row_index)
    a.out = _ptr(out)
    a.x_ss, a.x_sl = x.stride(0), x.stride(1)
    a.B_ss, a.B_sl = Bm.stride(0), Bm.stride(1)
    a.C_ss, a.C_sl = Cm.stride(0), Cm.stride(1)
    a.dt_sb, a.dt_sl = dt_tok.stride(0), dt_tok.stride(1)
    if z is not None:
        a.z_ss, a.z_sl = z.stride(0), z.stride(1)
    a.o_ss, a.o_sl = out.stride(0), out.stride(1)
    es = x.element_size()
    _launch("dm_ssd_fwd", a, x, (3 if z is not None else 2) * S * L * Din * es + 2 * S * L * 16 * es)
    return out


# Backward twin (csrc/ssd_bwd.hip): one workgroup per (sequence, head) recomputes the score tiles, nothing is saved by the
# forward (no delta tensor, no checkpoints).  DIFFMA_SSD_MFMA_BWD=0 keeps Mamba-2 training on the A-shared scan pair.
SSD_MFMA_BWD = os.environ.get("DIFFMA_SSD_MFMA_BWD", "1") == "1"


def ssd_bwd_supported(x, L, headdim, dstate, views=()):
    """The autograd node asks this before it commits the forward to the matrix pipe.  bf16 only: the kernel rounds its score and
    gradient tiles to the I/O dtype, and under fp16 autocast the GradScaler's 2^16 loss scale would push those tiles past the
    fp16 range (the scan pair keeps every intermediate in fp32) -- the fp16 instantiation exists and is parity-tested unscaled."""
    if not (SSD_MFMA and SSD_MFMA_BWD and x.is_cuda and x.dtype == torch.bfloat16):
        return False
    for v in views:
        if v is not None and (v.data_ptr() % 16 or v.stride(-1) != 1 or any(st % 8 for st in v.stride()[:-1])):
            return False
    return bool(_lib.load().dm_ssd_bwd_supported(int(L), int(headdim), int(dstate), dtype_code(x)))


def ssd_bwd(x, Bm, Cm, dt_tok, z, dout, A_h, D_h, dt_bias_h, *, z_row_index=None, out_row_index=None, batch_per_dir=0, dx_out=None):
    """Operands as ssd_fwd; dout: [S, L, H*64] gradient of the gated output, step l read at row out_row_index[dir][l].
    Returns (dx [S, L, H*64] scan order (dx_out view if given), dz [S, L, H*64] token order per direction,
    dBC fp32 [S, L, 32] (dB | dC, the per-head partial rows already summed), ddt fp32 [S, L, H] raw-dt gradient in token order per
    direction, dAD fp32 [3, H]: dA | dD | d dt_bias summed over the sequences)."""
    _require_gpu(x, Bm, Cm, dt_tok, z, dout)
    S, L, Din = x.shape
    H = A_h.shape[0]
    assert Din == H * 64 and Bm.shape[-1] == 16 and x.stride(2) == 1 and Bm.stride(2) == 1 and Cm.stride(2) == 1 and dt_tok.stride(2) == 1
    assert dout.shape == (S, L, Din) and dout.stride(2) == 1 and dout.dtype == x.dtype
    dx = dx_out if dx_out is not None else torch.empty((S, L, Din), dtype=x.dtype, device=x.device)
    dz = torch.empty((S, L, Din), dtype=x.dtype, device=x.device) if z is not None else None
    dbc = torch.empty((H, S, L, 32), dtype=torch.float32, device=x.device)       # head-major partial rows: summed by one column sum
    ddt = torch.empty((S, L, H), dtype=torch.float32, device=x.device)
    dad = torch.empty((S, 3, H), dtype=torch.float32, device=x.device)
    A32, D32, b32 = _f32c(A_h), _f32c(D_h), _f32c(dt_bias_h)
    a = dm_ssd_bwd_args()
    a.nseq, a.batch_per_dir, a.seqlen, a.nheads, a.headdim, a.dstate = S, batch_per_dir, L, H, 64, 16
    a.io_dtype, a.flags = dtype_code(x), 0
    a.x, a.B, a.C, a.dt, a.z, a.dout = _ptr(x), _ptr(Bm), _ptr(Cm), _ptr(dt_tok), _ptr(z), _ptr(dout)
    a.A, a.D, a.dt_bias = _ptr(A32), _ptr(D32), _ptr(b32)
    a.z_row_index, a.out_row_index = _ptr(z_row_index), _ptr(out_row_index)
    a.dx, a.dz = _ptr(dx), _ptr(dz)
    a.dBC_part, a.ddt, a.dAD_part = _ptr(dbc), _ptr(ddt), _ptr(dad)
    a.x_ss, a.x_sl = x.stride(0), x.stride(1)
    a.B_ss, a.B_sl = Bm.stride(0), Bm.stride(1)
    a.C_ss, a.C_sl = Cm.stride(0), Cm.stride(1)
    a.dt_sb, a.dt_sl = dt_tok.stride(0), dt_tok.stride(1)
    if z is not None:
        a.z_ss, a.z_sl = z.stride(0), z.stride(1)
        a.dz_ss, a.dz_sl = dz.stride(0), dz.stride(1)
    a.do_ss, a.do_sl = dout.stride(0), dout.stride(1)
    a.dx_ss, a.dx_sl = dx.stride(0), dx.stride(1)
    es = x.element_size()
    _launch("dm_ssd_bwd", a, x, (6 if z is not None else 3) * S * L * Din * es + 2 * S * L * 16 * es + S * H * L * 32 * 4)
    dbc_sum = colsum(dbc.view(H, S * L * 32)).view(S, L, 32) if H > 1 else dbc.view(S, L, 32)
    return dx, dz, dbc_sum, ddt, colsum(dad.view(S, 3 * H), True).view(3, H)
