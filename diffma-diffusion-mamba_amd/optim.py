"""AdamW + EMA of the weights in ONE pass over the parameters (csrc/optim.hip, `dm_adamw_ema_step`).

Reference: train.py:153-166 (`torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0)`), train.py:259-264 (`opt.step()`,
`update_ema(ema, model.module)`).  This is not a new optimiser: it runs the step of an existing `torch.optim.AdamW` instance ON THAT
INSTANCE'S STATE (`exp_avg`, `exp_avg_sq`, the per-parameter `step` tensors), so `optimizer.state_dict()` / `load_state_dict()` --
the checkpoint format -- are untouched and a run can switch between the two freely.  What changes is the traffic: torch's fused AdamW
(13 multi-tensor launches for DiffMa-L/2's 459 tensors, 28 bytes per element) + `_foreach_lerp_` for the EMA (8 launches, 12 bytes)
become one launch of 36 bytes per element plus a one-thread-per-tensor launch for the counters.  At one sample per GPU -- the
reference's own configuration -- the optimiser and the EMA are a sixth of the training step.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import dm_adamw_args

ENABLED = os.environ.get("DIFFMA_FUSED_OPT", "1") == "1"          # 0: torch's fused AdamW + _foreach_lerp_ again (A/B runs)


def supported(optimizer, params, ema_params=None) -> bool:
    """One AdamW parameter group without amsgrad / maximize / a tensor learning rate; fp32 contiguous parameters on one ROCm device."""
    if not ENABLED or type(optimizer) is not torch.optim.AdamW or len(optimizer.param_groups) != 1:
        return False
    g = optimizer.param_groups[0]
    if g.get("amsgrad") or g.get("maximize") or torch.is_tensor(g["lr"]) or g.get("differentiable"):
        return False
    ps = list(params)
    if not ps or any(p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.device != ps[0].device for p in ps):
        return False
    if ema_params is not None and any(e.dtype != torch.float32 or e.shape != p.shape or not e.is_contiguous() or e.device != p.device
                                      for e, p in zip(ema_params, ps)):
        return False
    return True


class FusedAdamWEMA:
    """step(found_inf=None): one AdamW step of `optimizer` on `params` (the tensors with a gradient) and, with `ema_params`, the EMA
    step towards the new weights, in one pass.  The tensor table is rebuilt when a pointer moved (eager training re-allocates the
    gradients every step; inside a captured graph everything is static and the table is built once, at capture)."""

    def __init__(self, optimizer, params, ema_params=None, ema_decay=0.9999, ema_on_skip=True):
        self.opt = optimizer
        self.params = list(params)
        self.ema = list(ema_params) if ema_params is not None else None
        if self.ema is not None and len(self.ema) != len(self.params):
            raise ValueError("ema_params must pair up with params")
        self.decay, self.ema_on_skip = float(ema_decay), bool(ema_on_skip)
        self.chunk = int(_lib.load().dm_adamw_chunk())
        self._key, self._table, self._bt, self._bc, self._host, self._sizes = None, None, None, None, None, None
        dev = self.params[0].device
        for p in self.params:                                     # the state torch's own step would create lazily (fused / capturable layout)
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
                st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=dev)        # a CPU counter of an unfused run

    def _build(self, live):
        rows = []
        for i in live:
            p, st = self.params[i], self.opt.state[self.params[i]]
            rows.append((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.grad.data_ptr(),
                         self.ema[i].data_ptr() if self.ema is not None else 0, st["step"].data_ptr(), p.numel()))
        key = tuple(rows)
        if key == self._key:
            return
        dev = self.params[0].device
        if self._host is None:                                    # allocated once, outside any capture: 7 x 8 bytes per row = dm_adamw_tensor
            self._host = torch.empty((len(self.params), 7), dtype=torch.int64).pin_memory()
            self._table = torch.empty((len(self.params), 7), dtype=torch.int64, device=dev)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            torch.cuda.current_stream(dev).synchronize()          # an earlier asynchronous copy may still be reading the pinned rows
        self._host[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
        # captured: a copy node that replays from the pinned rows (this object then belongs to that graph: no eager steps through it)
        self._table.copy_(self._host, non_blocking=True)
        sizes = tuple(r[6] for r in rows)
        if self._bt is None or self._sizes != sizes:
            if capturing:
                raise RuntimeError("FusedAdamWEMA: the set of parameters with gradients changed inside a capture; run one eager step first")
            bt, bc = [], []
            for t, n in enumerate(sizes):
                nb = (n + self.chunk - 1) // self.chunk
                bt += [t] * nb
                bc += list(range(nb))
            self._bt = torch.tensor(bt, dtype=torch.int32, device=dev)
            self._bc = torch.tensor(bc, dtype=torch.int32, device=dev)
            self._sizes = sizes
        self._key = key

    def _live(self):
        live = [i for i, p in enumerate(self.params) if p.grad is not None]
        for i in live:
            g = self.params[i].grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise RuntimeError("FusedAdamWEMA needs fp32 contiguous gradients")
        return live

    @torch.no_grad()
    def nonfinite(self, found):
        """found (0-dim fp32 device tensor, zeroed by the caller) becomes 1 if any gradient holds an Inf / NaN: one launch over the table."""
        live = self._live()
        if not live:
            return found
        self._build(live)
        a = dm_adamw_args()
        a.tensors, a.block_tensor, a.block_chunk = self._table.data_ptr(), self._bt.data_ptr(), self._bc.data_ptr()
        a.ntensors, a.nblocks = len(live), int(self._bt.numel())
        a.nonfinite_out = found.data_ptr()
        dev = self.params[0].device
        with torch.cuda.device(dev):
            _lib.call("dm_grads_nonfinite", a, torch.cuda.current_stream(dev).cuda_stream)
        return found

    @torch.no_grad()
    def step(self, found_inf=None):
        live = self._live()
        if not live:
            return
        self._build(live)
        grp = self.opt.param_groups[0]
        a = dm_adamw_args()
        a.tensors, a.block_tensor, a.block_chunk = self._table.data_ptr(), self._bt.data_ptr(), self._bc.data_ptr()
        a.ntensors, a.nblocks = len(live), int(self._bt.numel())
        a.lr, (a.beta1, a.beta2), a.eps, a.weight_decay = float(grp["lr"]), grp["betas"], float(grp["eps"]), float(grp["weight_decay"])
        a.ema_decay = self.decay
        a.found_inf = found_inf.data_ptr() if found_inf is not None else 0
        a.ema_on_skip = 1 if self.ema_on_skip else 0
        dev = self.params[0].device
        with torch.cuda.device(dev):
            _lib.call("dm_adamw_ema_step", a, torch.cuda.current_stream(dev).cuda_stream)
        self._keep = (a, found_inf)
