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
_GEN = [0]               # bumped by invalidate(): weights changed behind autograd's back (a hipGraph replay of the optimizer, `.data` writes)
_SHADOWS = {}            # id(master) -> (weakref to the master, shadow tensor, master version when copied, _GEN then); the weakref guards against
                         # id() reuse after a model is freed (a NEW parameter with the id and version of a dead one must not get its shadow)
                         # and its callback drops the entry (and with it the shadow's device memory) when the master dies


def _register(w, shadow):
    key = id(w)

    def _gone(ref, key=key):
        ent = _SHADOWS.get(key)
        if ent is not None and ent[0] is ref:
            del _SHADOWS[key]

    _SHADOWS[key] = (weakref.ref(w, _gone), shadow, w._version, _GEN[0])


def invalidate():
    """Every 16-bit copy made so far is stale from now on.  For writers that do not bump `_version`: a replayed hipGraph of the fused
    AdamW (graphed.GraphedTrainStep.step calls this next to dropping `_A_cache`), `.data` assignments.  The next eager prepare() then
    re-casts the masters and cast_weight() falls back to a fresh cast until it has (before this an eager grad-enabled
    forward after a replay multiplied by weight copies one optimizer step old)."""
    _GEN[0] += 1
    from .block_ops import drop_mask_cache
    drop_mask_cache()


class _NegExpAll(torch.autograd.Function):
    """(-exp(a_1), ..., -exp(a_n)) with two foreach launches; backward: grad_i * out_i in one."""

    @staticmethod
    def forward(ctx, *logs):
        ctx.set_materialize_grads(False)       # a staged backward (graphed.StagedBackward) asks for two of the outputs' gradients at a time:
                                               # without this autograd would fill a zero tensor for each of the others on every call
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
    if ent[3] == _GEN[0] and ent[2] == weight._version and ent[1].dtype == dtype and ent[1].device == weight.device and ent[1].shape == weight.shape:
        return ent[1]
    return None


def cast_weight(weight, dtype):
    """weight in `dtype`: the step's shadow when it is current, a fresh cast otherwise."""
    if weight.dtype == dtype:
        return weight
    s = shadow_of(weight, dtype)
    return s if s is not None else weight.to(dtype)


def stacked_pair(w0, w1, dtype):
    """[2, ...] tensor of the two weights in `dtype`: the shared buffer of their shadows when both are current (no launch),
    else a stack of the casts."""
    s0, s1 = shadow_of(w0, dtype) if w0.dtype != dtype else None, shadow_of(w1, dtype) if w1.dtype != dtype else None
    if s0 is not None and s1 is not None:
        base = s0._base
        if (base is not None and base is s1._base and base.dim() == s0.dim() + 1 and base.shape[0] == 2
                and s0.data_ptr() == base.data_ptr() and s1.data_ptr() == base.data_ptr() + s0.numel() * s0.element_size()):
            return base
    return torch.stack([cast_weight(w0, dtype), cast_weight(w1, dtype)])


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
                ada = getattr(m, "adaLN_modulation", None)          # the block's adaLN Linear goes through linear_splitk too (mamba_block.py)
                if isinstance(ada, torch.nn.Sequential) and isinstance(ada[1], torch.nn.Linear) and ada[1].bias is not None:
                    masters += [ada[1].weight, ada[1].bias]
        # weights of two consecutive mixers with equal shapes (mamba1 / mamba2 of a Spiral_MambaBlock): their 16-bit copies are
        # the two halves of ONE [2, ...] buffer, which the paired-mixer path hands to a batched GEMM as it is (stacked_pair)
        names = ("in_proj", "out_proj", "x_proj", "dt_proj")
        mods = [m for m in model.modules() if hasattr(m, "A_log") and all(isinstance(getattr(m, n, None), torch.nn.Linear) for n in names)]
        partner = {}
        for a, b in zip(mods[0::2], mods[1::2]):
            for n in names:
                wa, wb = getattr(a, n).weight, getattr(b, n).weight
                if wa.shape == wb.shape and wa.dtype == wb.dtype:
                    partner[id(wa)] = wb
        # the blocks' adaLN Linear layers all read the same SiLU(c): their 16-bit copies are the rows of ONE [nblocks, 3 D, 2 D] buffer
        # (and one [nblocks, 3 D] buffer for the biases), which `adaln_all` multiplies in one product instead of one per block
        ada_w, ada_b = [], []
        for m in getattr(model, "blocks", []):
            ada = getattr(m, "adaLN_modulation", None)
            if getattr(m, "attention_network", None) is not None and isinstance(ada, torch.nn.Sequential) and isinstance(ada[1], torch.nn.Linear) \
                    and ada[1].bias is not None:
                ada_w.append(ada[1].weight)
                ada_b.append(ada[1].bias)
        if len(ada_w) < 2 or any(w.shape != ada_w[0].shape or w.dtype != ada_w[0].dtype for w in ada_w):
            ada_w, ada_b = [], []
        plan = _PLANS[model] = dict(mixers=mixers, masters=masters, shadows={}, partner=partner, ada_w=ada_w, ada_b=ada_b, ada_stack={})
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
            fresh = any(s._use_count() > 1 for s in sh) or any(b._use_count() > n for b, n in plan.get("bases", {}).get(dt, ()))
        if fresh:
            sh, made, bases = [], {}, []
            if plan["ada_w"]:                                      # rows of one buffer each, in block order
                nb = len(plan["ada_w"])
                wbase = torch.empty((nb,) + tuple(plan["ada_w"][0].shape), dtype=dt, device=masters[0].device)
                bbase = torch.empty((nb,) + tuple(plan["ada_b"][0].shape), dtype=dt, device=masters[0].device)
                for i, (w, b) in enumerate(zip(plan["ada_w"], plan["ada_b"])):
                    made[id(w)], made[id(b)] = wbase[i], bbase[i]
                plan["ada_stack"][dt] = (wbase, bbase)
                bases += [wbase, bbase]                            # what adaln_all's autograd node saves: part of the live-graph check below
            for w in masters:
                if id(w) in made:
                    sh.append(made.pop(id(w)))
                    continue
                wb = plan["partner"].get(id(w))
                if wb is not None:
                    base = torch.empty((2,) + tuple(w.shape), dtype=dt, device=w.device)
                    sh.append(base[0])
                    made[id(wb)] = base[1]
                    bases.append(base)
                else:
                    sh.append(torch.empty_like(w, dtype=dt))
            plan["shadows"][dt] = sh
            # (a [2, ...] base handed out by stacked_pair may be what an autograd node saved: its reference count at rest
            #  -- the plan's own reference and its two views -- is the baseline the live-graph check compares with)
            plan.setdefault("bases", {})[dt] = [(b, b._use_count()) for b in bases]
        with torch.no_grad():
            torch._foreach_copy_(sh, [w.detach() for w in masters])
        for w, s in zip(masters, sh):
            _register(w, s)


def adaln_stack(model, dtype):
    """([nblocks, 3 D, 2 D] weight, [nblocks, 3 D] bias) in `dtype`: the blocks' adaLN Linear layers as ONE operand, valid when every
    row is the current shadow of its master (prepare() ran this step and nothing was written since); None otherwise."""
    plan = _PLANS.get(model)
    if plan is None or not plan.get("ada_w"):
        return None
    st = plan["ada_stack"].get(dtype)
    if st is None:
        return None
    wbase, bbase = st
    for i, (w, b) in enumerate(zip(plan["ada_w"], plan["ada_b"])):
        sw, sb = shadow_of(w, dtype), shadow_of(b, dtype)
        if sw is None or sb is None or sw.data_ptr() != wbase[i].data_ptr() or sb.data_ptr() != bbase[i].data_ptr():
            return None
    return wbase, bbase                                        # the BASES ([nblocks, 3 D, 2 D], [nblocks, 3 D]): an autograd node must save these objects


def adaln_params(model):
    plan = _PLANS.get(model)
    return (plan["ada_w"], plan["ada_b"]) if plan is not None and plan.get("ada_w") else None
