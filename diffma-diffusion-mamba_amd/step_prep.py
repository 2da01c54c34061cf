"""Once-per-step preparation of small per-module tensors in a handful of multi-tensor launches.

A DiffMa-L/2 training step is ~2 300-2 800 kernel launches; at the reference's own batch (config/brain.yaml: 1 sample per GPU) the
step is bound by the launch rate (eager) or by per-node dispatch (hipGraph replay).  Two families of tiny launches repeat per mixer:
  * `A = -exp(A_log)` (exp, neg; mul, neg in the backward)         -- 32 mixers x ~5 launches
  * the autocast copies of the projection weights (in_proj, x_proj, dt_proj, out_proj, the fusion MLP)   -- ~160 cast launches
`prepare(model)` does both for the WHOLE model at the top of `DiffMa.forward` with foreach kernels: the mixers pick their `A` from
`_A_step`, and `shadow_of(weight, dtype)` hands the 16-bit copy to `_LinearSplitKFn` / `_SpiralSSMFn` as long as the master has not
been written since (its `_version` is compared: a stale or unknown weight falls back to the per-call cast, so nothing depends on
`prepare` having run).  Inside a captured training step the foreach launches are part of the graph and run on every replay.
"""
from __future__ import annotations

import os
import weakref

import torch

ENABLED = os.environ.get("DIFFMA_STEP_PREP", "1") == "1"
_PLANS = weakref.WeakKeyDictionary()
_SHADOWS = {}            # id(master) -> (weakref to the master, shadow tensor, master version when copied); the weakref guards against
                         # id() reuse after a model is freed (a NEW parameter with the id and version of a dead one must not get its shadow)
                         # and its callback drops the entry (and with it the shadow's device memory) when the master dies


def _register(w, shadow):
    key = id(w)

    def _gone(ref, key=key):
        ent = _SHADOWS.get(key)
        if ent is not None and ent[0] is ref:
            del _SHADOWS[key]

    _SHADOWS[key] = (weakref.ref(w, _gone), shadow, w._version)


class _NegExpAll(torch.autograd.Function):
    """(-exp(a_1), ..., -exp(a_n)) with two foreach launches; backward: grad_i * out_i in one."""

    @staticmethod
    def forward(ctx, *logs):
        outs = torch._foreach_exp([a.float() if a.dtype != torch.float32 else a for a in logs])
        torch._foreach_neg_(outs)
        ctx.save_for_backward(*outs)
        ctx.dtypes = [a.dtype for a in logs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        outs = ctx.saved_tensors
        idx = [i for i, g in enumerate(grads) if g is not None]
        res = [None] * len(outs)
        if idx:
            prod = torch._foreach_mul([grads[i] for i in idx], [outs[i] for i in idx])
            for i, p in zip(idx, prod):
                res[i] = p if p.dtype == ctx.dtypes[i] else p.to(ctx.dtypes[i])
        return tuple(res)


def shadow_of(weight, dtype):
    """The 16-bit copy made by the last prepare() if the master is unchanged since, else None."""
    ent = _SHADOWS.get(id(weight))
    if ent is None:
        return None
    if ent[0]() is not weight:                      # the entry belongs to a parameter that no longer exists
        del _SHADOWS[id(weight)]
        return None
    if ent[2] == weight._version and ent[1].dtype == dtype and ent[1].device == weight.device and ent[1].shape == weight.shape:
        return ent[1]
    return None


def cast_weight(weight, dtype):
    """weight in `dtype`: the step's shadow when it is current, a fresh cast otherwise."""
    if weight.dtype == dtype:
        return weight
    s = shadow_of(weight, dtype)
    return s if s is not None else weight.to(dtype)


def prepare(model, dtype=None):
    """Called at the top of DiffMa.forward when gradients are on and the model is on a ROCm device.  dtype: the 16-bit dtype of the
    weight copies (default: the active CUDA autocast dtype; tests pass it explicitly)."""
    if not ENABLED:
        return
    plan = _PLANS.get(model)                        # per model, outside its __dict__: deepcopy / state_dict never see it
    if plan is None:
        from .mamba import Mamba
        mixers = [m for m in model.modules() if isinstance(m, Mamba)]
        masters = []
        for m in model.modules():
            if hasattr(m, "A_log"):                                    # a Mamba / Mamba2 mixer
                for name in ("in_proj", "out_proj", "x_proj", "dt_proj"):
                    lin = getattr(m, name, None)
                    if isinstance(lin, torch.nn.Linear):
                        masters.append(lin.weight)
            net = getattr(m, "attention_network", None)
            if net is not None:
                masters += [net[1].weight, net[3].weight]
        plan = _PLANS[model] = dict(mixers=mixers, masters=masters, shadows={})
    mixers = plan["mixers"]
    if mixers:
        As = _NegExpAll.apply(*[m.A_log for m in mixers])
        for m, a in zip(mixers, As):
            m.__dict__["_A_step"] = a
    dt = dtype if dtype is not None else (torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None)
    if dt in (torch.bfloat16, torch.float16) and plan["masters"]:
        masters = plan["masters"]
        sh = plan["shadows"].get(dt)
        fresh = sh is None or sh[0].device != masters[0].device
        capturing = masters[0].is_cuda and torch.cuda.is_current_stream_capturing()
        if not fresh and not capturing:
            if all(shadow_of(w, dt) is s for w, s in zip(masters, sh)):
                # Every shadow is current (no master written since the last prepare): nothing to copy.  This is also what keeps a
                # SECOND forward before the first one's backward legal: the shadows are saved for backward by the projections'
                # autograd nodes, and an in-place refresh would invalidate the first graph.
                return
            # Masters were written (an optimizer step).  If an autograd graph still holds the old copies (retain_graph, a backward
            # that has not run yet) they must not be overwritten: new buffers, the old ones die with that graph.
            fresh = any(s._use_count() > 1 for s in sh)
        if fresh:
            sh = plan["shadows"][dt] = [torch.empty_like(w, dtype=dt) for w in masters]
        with torch.no_grad():
            torch._foreach_copy_(sh, [w.detach() for w in masters])
        for w, s in zip(masters, sh):
            _register(w, s)
