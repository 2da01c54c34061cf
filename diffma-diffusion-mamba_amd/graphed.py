"""hipGraph capture of the denoiser step for sampling.

A DiffMa forward at batch 1-8 is ~400 kernel launches of a few microseconds each: launch-bound, exactly the
regime SURVEY.md 7.3 describes.  Shapes are static across all 250 (or 50) sampling steps, every C-ABI launch is
asynchronous and allocation-free, and the diffusion tables live on the device, so the whole
`model(x, t, y, y2, w)` call is captured ONCE into a HIP graph (torch.cuda.CUDAGraph drives hipGraph on ROCm)
and replayed per step with the inputs copied into static buffers.
"""
from __future__ import annotations

import torch


def _capture_mode(device):
    """With a process group alive (multi-rank sampling / training), RCCL's watchdog thread polls its work events (hipEventQuery) at
    any time; under the default "global" capture mode such a call from ANOTHER thread invalidates this thread's capture
    ("operation not permitted when stream is capturing", seen as an abort of `bench.py --graph` under torch.distributed.run).  Let
    the collectives issued so far retire, then capture in thread-local mode."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        torch.cuda.synchronize(device)
        return "thread_local"
    return "global"


class GraphedDenoiser:
    """Callable with the model's signature `(x, t, y=, y2=, w=)`; replays a captured graph.

    The conditioning tensors y, y2, w are fixed for a whole sampling run: they are copied into the static
    buffers only when their storage changes.

    With an autocast dtype the graph runs on a SNAPSHOT of the model whose nn.Linear weight matrices are stored in
    that dtype (autocast would otherwise re-cast every fp32 weight inside the captured graph on every replay: ~180
    cast kernels per step).  Everything autocast keeps in fp32 (norm weights, biases, A_log, D, the residual stream)
    stays fp32, so the outputs are those of the autocast model.  Pass `snapshot_weights=False` to capture the live
    model instead (e.g. when its weights keep changing)."""

    def __init__(self, model, x, t, y, y2, w, autocast_dtype=None, warmup=3, snapshot_weights=True):
        assert x.is_cuda, "graph capture needs a ROCm device"
        if autocast_dtype is not None and snapshot_weights:
            import copy
            model = copy.deepcopy(model).eval()
            for m in model.modules():
                if isinstance(m, torch.nn.Linear):
                    m.weight.data = m.weight.data.to(autocast_dtype)
        self.model = model
        self.amp = autocast_dtype
        self.sx, self.st = x.clone(), t.clone()
        self.sy, self.sy2, self.sw = y.clone(), y2.clone(), w.clone()
        self._cond_id = None
        # everything below runs on the INPUT's device: torch.cuda.Stream() / torch.cuda.graph() use the current device, and
        # a capture on cuda:0 of a model living on cuda:k records nothing (every replay would return the warm-up output)
        with torch.cuda.device(x.device):
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream(x.device))
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):                   # lazy inits (hipBLASLt heuristics, allocator) happen outside the capture
                    self._run()
            torch.cuda.current_stream(x.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=_capture_mode(x.device)):
                self.sout = self._run()

    def _run(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            return self.model(self.sx, self.st, y=self.sy, y2=self.sy2, w=self.sw)

    def set_condition(self, y, y2, w):
        """Copy the conditioning of a sampling run into the static buffers (also done by __call__ when it sees new tensors)."""
        self.sy.copy_(y)
        self.sy2.copy_(y2)
        self.sw.copy_(w)
        self._cond_id = tuple((t.data_ptr(), t._version) for t in (y, y2, w))

    def __call__(self, x, t, y=None, y2=None, w=None, **kw):
        self.sx.copy_(x)
        self.st.copy_(t)
        # storage AND version: a caller that refills the same buffers in place bumps _version
        if tuple((c.data_ptr(), c._version) for c in (y, y2, w)) != self._cond_id:
            self.set_condition(y, y2, w)
        with torch.cuda.device(self.sx.device):
            self.graph.replay()
        return self.sout

    def parameters(self):
        return self.model.parameters()


class StagedBackward:
    """The backward of one DiffMa training loss in STAGES of `per` blocks, last blocks first, so that the gradients of a stage are
    complete -- and can be all-reduced -- while the earlier blocks' backward still runs (GraphedTrainStep captures one hipGraph per
    stage and launches each stage's all-reduce between replays; reference train.py:153 overlaps the same way through DDP's
    buckets, which a captured backward cannot use).

    autograd has no "stop here" for an interior tensor, so the cuts are made in the FORWARD: while `with staged:` is active, hooks
    replace every block's output (and the conditioning vector c, and the embedded input of block 0) by a detached leaf that all
    later consumers read -- the next block, and the long-skip partner (reference model.py:286-295: block i > depth/2 reads
    outs[i-1] + outs[depth-1-i]).  The backward is then one `torch.autograd.backward(block output, grad_tensors=leaf.grad,
    inputs=block parameters + the leaves the block read)` per block, last block first: by the time block j runs, every consumer of
    its output has deposited its gradient in the leaf.  Parameter gradients accumulate in `.grad` exactly as with one `backward()`;
    the arithmetic is the same, only the order in which the contributions to c (16 blocks + head) are added differs."""

    def __init__(self, model, per=4):
        self.model, self.depth, self.per = model, len(model.blocks), per
        if self.depth < 2 * per or self.depth % per:
            raise ValueError(f"staged backward needs depth ({self.depth}) to be a multiple of {per} and at least {2 * per}")
        self.nstage = self.depth // per
        blk_params = [[p for p in b.parameters() if p.requires_grad] for b in model.blocks]
        in_blocks = {id(p) for ps in blk_params for p in ps}
        rest = [p for p in model.parameters() if p.requires_grad and id(p) not in in_blocks]
        tail = {id(p) for p in model.final_layer.parameters()}
        self.blk_params = blk_params
        self.head_params = [p for p in rest if id(p) in tail]
        self.embed_params = [p for p in rest if id(p) not in tail]
        self.params = []                                   # per stage, in the order the stages run (last blocks first)
        for s in range(self.nstage - 1, -1, -1):
            ps = [p for b in range(s * per, (s + 1) * per) for p in blk_params[b]]
            if s == self.nstage - 1:
                ps = ps + self.head_params
            if s == 0:
                ps = ps + self.embed_params
            self.params.append(ps)
        self._hooks = []

    # ---- forward-side cuts ------------------------------------------------------------------------------------------------
    def __enter__(self):
        self.model._no_adaln_all = True            # the cuts below re-route `c` per block: each block runs its own adaLN product
        self.orig, self.leaf = [None] * self.depth, [None] * self.depth
        self.c_orig = self.c_leaf = self.x0_orig = self.x0_leaf = None
        cut = lambda t: t.detach().requires_grad_(True)

        def pre(k):
            def hook(m, args):
                x, c = args[0], args[1]
                if k == 0:
                    self.c_orig, self.c_leaf = c, cut(c)
                    self.x0_orig, self.x0_leaf = x, cut(x)
                    x = self.x0_leaf
                return (x, self.c_leaf) + tuple(args[2:])
            return hook

        def post(k):
            def hook(m, args, out):
                self.orig[k], self.leaf[k] = out, cut(out)
                return self.leaf[k]
            return hook

        for k, b in enumerate(self.model.blocks):
            self._hooks.append(b.register_forward_pre_hook(pre(k)))
            self._hooks.append(b.register_forward_hook(post(k)))
        self._hooks.append(self.model.final_layer.register_forward_pre_hook(lambda m, args: (args[0], self.c_leaf) + tuple(args[2:])))
        return self

    def __exit__(self, *exc):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.model._no_adaln_all = False
        return False

    def _reads(self, i):
        """the leaves block i read: the chain activation and, past the middle of the stack, its long-skip partner"""
        ins = [self.x0_leaf] if i == 0 else [self.leaf[i - 1]]
        if i > self.depth / 2:
            ins.append(self.leaf[self.depth - i - 1])
        return ins

    # ---- backward, stage by stage -----------------------------------------------------------------------------------------
    def run_stage(self, k, loss=None):
        """Stage k (0 = the LAST `per` blocks + the head; nstage - 1 = the first blocks + the embedders): afterwards the `.grad` of
        self.params[k] is complete."""
        s = self.nstage - 1 - k
        if k == 0:
            torch.autograd.backward(loss, inputs=self.head_params + [self.leaf[self.depth - 1], self.c_leaf], retain_graph=True)
        for i in range((s + 1) * self.per - 1, s * self.per - 1, -1):
            g = self.leaf[i].grad
            if g is None:                                   # an output nobody consumed (cannot happen in DiffMa's wiring)
                continue
            torch.autograd.backward(self.orig[i], grad_tensors=g, inputs=self.blk_params[i] + self._reads(i) + [self.c_leaf], retain_graph=True)
        if s == 0:
            outs_, gs = [self.x0_orig, self.c_orig], [self.x0_leaf.grad, self.c_leaf.grad]
            keep = [(o, g) for o, g in zip(outs_, gs) if g is not None and o.requires_grad]
            if keep and self.embed_params:
                torch.autograd.backward([o for o, _ in keep], grad_tensors=[g for _, g in keep], inputs=self.embed_params)

    def run(self, loss, on_stage=None):
        for k in range(self.nstage):
            self.run_stage(k, loss)
            if on_stage is not None:
                on_stage(k)


class GraphedTrainStep:
    """One whole optimisation step -- q_sample noise, denoiser forward, loss, backward, fused AdamW, EMA -- captured in
    ONE hipGraph and replayed per iteration.

    A DiffMa-L/2 step is ~2 900 kernel launches; below ~90 samples per GPU the host cannot issue them as fast as the GPU
    retires them (55 ms/step floor measured, e.g. at the reference's own `global_batch_size: 8`).  Shapes are static
    across steps, the C-ABI launches are asynchronous and allocation-free and the diffusion tables live on the device,
    so the step is captured once (after eager warm-up iterations that also settle the GEMM table) and replayed with the
    batch copied into static buffers.  The noise of `training_losses` is drawn inside the graph (PyTorch's graph-safe
    Philox offsets advance per replay).  Construction leaves weights, EMA and optimizer state as it found them.

    Data parallel (`process_group` with more than one rank, or `split=True`): the step is TWO graphs with the gradient
    all-reduce between them -- graph 1 = forward + backward + the gradients flattened into one buffer, then ONE eager
    `all_reduce(AVG)` over RCCL (nothing of the collective is captured), graph 2 = gradients back into place, fused AdamW, EMA.
    This is the reference's own `brain.yaml` case (global batch 8 on 8 GPUs = ONE sample per GPU), where the eager DDP step is
    bound by the host's launch rate on every rank; the all-reduce is not overlapped with the backward (it follows a ~10-20 ms
    replay instead of hiding in a 55 ms host-bound step).  The caller keeps the ranks' weights identical at entry (train.py
    broadcasts them) and does NOT wrap the model in DistributedDataParallel.

    step(z, t, y, y2, w) -> loss (a 0-d device tensor that is overwritten by the next replay; the local rank's loss).
    """

    def __init__(self, model, ema, optimizer, diffusion, z, t, y, y2, w, autocast_dtype=None, ema_decay=0.9999, warmup=3,
                 process_group=None, split=None, stages=None):
        assert z.is_cuda, "graph capture needs a ROCm device"
        for g in optimizer.param_groups:
            if not g.get("capturable", False):
                raise ValueError("the optimizer must be built with capturable=True (and fused=True) to be captured")
        self.model, self.ema, self.opt, self.diffusion = model, ema, optimizer, diffusion
        self.amp, self.decay = autocast_dtype, ema_decay
        self.sz, self.st = z.clone(), t.clone()
        self.sy, self.sy2, self.sw = y.clone(), y2.clone(), w.clone()
        self._mixers = [m for net in (model, ema) if net is not None for m in net.modules() if hasattr(m, "A_log")]
        self._ep = [p for p in ema.parameters()] if ema is not None else []
        self._mp = [p for p in model.parameters()] if ema is not None else []
        # the warm-up iterations are real optimisation steps: remember the training state and put it back afterwards, so
        # that constructing this object does not advance the run
        with torch.no_grad():
            p_snap = [p.detach().clone() for p in model.parameters()]
            e_snap = [p.detach().clone() for p in self._ep]
        import torch.distributed as dist
        self.pg = process_group
        nranks = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.split = bool(split) if split is not None else nranks > 1
        if self.split and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("the two-graph (data-parallel) step needs an initialised process group")
        # Data-parallel form in STAGES (opt-in: stages=4 / DIFFMA_GRAPH_STAGES=4): the backward is cut into groups of blocks
        # (StagedBackward), one hipGraph per group, and each group's gradients are all-reduced -- asynchronously, on RCCL's stream --
        # while the next group's graph replays.  Default 0 = ONE all-reduce after the whole backward: on one rank the staged form
        # costs +1.15 ms at one sample per GPU and +0.4 ms at batch 8 (per-block autograd calls re-walk the nodes the blocks share,
        # three more graph launches and collectives), about what hiding three quarters of a ~0.7 ms all-reduce would return on
        # 8 GPUs -- it has to be measured there before it becomes the default.
        import os
        if stages is None:
            stages = int(os.environ.get("DIFFMA_GRAPH_STAGES", "0"))
        self.staged = None
        if self.split and stages and hasattr(model, "blocks") and hasattr(model, "final_layer") and len(model.blocks) >= 2 * 4 \
                and len(model.blocks) % 4 == 0 and type(model.blocks[0]).__name__ == "Spiral_MambaBlock":
            # the stage count is rounded DOWN to a divisor of the depth (depth 16, stages 3 -> 2 stages of 8 blocks); without one
            # (>= 2 stages of >= 4 blocks) the step stays in the two-graph form instead of raising at construction
            depth = len(model.blocks)
            want = max(2, min(int(stages), depth // 4))
            nst = next((k for k in range(want, 1, -1) if depth % k == 0 and depth // k >= 4), 0)
            if nst:
                self.staged = StagedBackward(model, per=depth // nst)
        self._gp = [p for p in model.parameters() if p.requires_grad]
        # AdamW + EMA in one pass over the parameters (csrc/optim.hip) on the optimizer's own state, when it is the plain AdamW the
        # reference trains with (train.py:153-166); the parameters without a gradient keep their EMA through the multi-tensor lerp
        from . import optim as _optim
        self._fused = None
        ema_of = {id(p): e for p, e in zip(self._mp, self._ep)} if ema is not None else None
        if _optim.supported(optimizer, self._gp, [ema_of[id(p)] for p in self._gp] if ema_of else None):
            self._fused = _optim.FusedAdamWEMA(optimizer, self._gp, [ema_of[id(p)] for p in self._gp] if ema_of else None,
                                               ema_decay=ema_decay, ema_on_skip=True)
            live = {id(p) for p in self._gp}
            self._ep_rest = [e for p, e in zip(self._mp, self._ep) if id(p) not in live]
            self._mp_rest = [p for p in self._mp if id(p) not in live]
        # taken AFTER FusedAdamWEMA's constructor: it replaces a CPU / non-fp32 `step` counter of an optimizer that has already stepped
        # by a device tensor of the same value -- a snapshot keyed by id() from before would miss the new tensor and the restore
        # below would zero the bias-correction count
        with torch.no_grad():
            o_snap = {id(v): v.detach().clone() for st in optimizer.state.values() for v in st.values() if torch.is_tensor(v)}
        # Non-finite gradients (the reference's eager loop skips such a step, train.py:254-256): a replayed graph cannot branch on
        # the host, so the decision is taken ON THE DEVICE -- found_inf is computed from the (all-reduced, hence rank-identical)
        # gradients inside the graph and handed to the fused AdamW, which then leaves weights, moments and step count alone.
        # `skipped` counts such steps; the host reads it when it logs (and does not count them as optimisation steps).
        self._one = torch.ones((), device=z.device)
        self.skipped = torch.zeros((), device=z.device)
        with torch.cuda.device(z.device):
            side = torch.cuda.Stream(device=z.device)
            side.wait_stream(torch.cuda.current_stream(z.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):                   # lazy inits, GEMM solution lookups, optimizer state allocation
                    if self.staged is not None:
                        self._staged_eager()
                    elif self.split:
                        self._fwd_bwd()
                        self._all_reduce()
                        self._update()
                    else:
                        self._step()
            torch.cuda.current_stream(z.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            mode = _capture_mode(z.device)
            if self.staged is not None:
                self.stage_graphs, self.flats = [], []
                for k in range(self.staged.nstage):
                    g = self.graph if k == 0 else torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=mode, **({} if k == 0 else {"pool": self.graph.pool()})):
                        if k == 0:
                            self.sloss = self._staged_forward()
                        self.flats.append(self._staged_stage(k))
                    self.stage_graphs.append(g)
                self.graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode=mode):
                    self._update()
            elif self.split:
                with torch.cuda.graph(self.graph, capture_error_mode=mode):       # gradients are allocated inside the graph's pool and stay attached
                    self.sloss = self._fwd_bwd()
                self.graph2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode=mode):
                    self._update()
            else:
                with torch.cuda.graph(self.graph, capture_error_mode=mode):
                    self.sloss = self._step()
        with torch.no_grad():
            for p, q in zip(model.parameters(), p_snap):
                p.copy_(q)
            for p, q in zip(self._ep, e_snap):
                p.copy_(q)
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.copy_(o_snap[id(v)]) if id(v) in o_snap else v.zero_()
            self.skipped.zero_()        # the warm-up iterations ran on restored state: their dropped steps are not the run's

    def _step(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            loss = self.diffusion.training_losses(self.model, self.sz, self.st, dict(y=self.sy, y2=self.sy2, w=self.sw))["loss"].mean()
        loss.backward()
        found = torch.zeros((), device=self.sz.device)            # 0-dim like GradScaler's (the fused AdamW subtracts it from its 0-dim step counters)
        grads = [p.grad for p in self._gp if p.grad is not None]
        if self._fused is not None and len(grads) == len(self._gp):
            self._fused.nonfinite(found)                                                    # one launch over the optimiser's tensor table
        elif grads:
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found, self._one)      # inv_scale 1: a pure check
        self._guarded_update(found)
        self.opt.zero_grad(set_to_none=True)
        return loss.detach()

    def _guarded_update(self, found):
        if self._fused is not None and all(p.grad is not None for p in self._gp):
            self._fused.step(found)                                   # weights, moments, counters and the EMA in one pass; found = 1 -> EMA only
            with torch.no_grad():
                self.skipped += found
                if self._ep_rest:
                    torch._foreach_lerp_(self._ep_rest, self._mp_rest, 1.0 - self.decay)
            return
        self.opt.grad_scale, self.opt.found_inf = None, found          # read by the fused AdamW: found_inf = 1 -> no update
        self.opt.step()
        with torch.no_grad():
            self.skipped += found
            if self.ema is not None:
                # ONE multi-tensor pass (ema += (1 - decay) (model - ema): 3 accesses per element; mul_ + add_ are 5, and a blend
                # gated by `found` on the device needs 8 -- at one sample per GPU the EMA and AdamW passes over the 89 M parameters
                # are 15 % of the step).  On a dropped step the weights did not move, so the EMA takes one ordinary step towards
                # them; the reference `continue`s before update_ema (train.py:254-264) -- a 1 - decay difference on such a step.
                torch._foreach_lerp_(self._ep, self._mp, 1.0 - self.decay)

    # ---- the data-parallel form: graph 1 | all-reduce | graph 2 ----------------------------------------------------------
    def _fwd_bwd(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            loss = self.diffusion.training_losses(self.model, self.sz, self.st, dict(y=self.sy, y2=self.sy2, w=self.sw))["loss"].mean()
        loss.backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self._gp]
        self.flat = torch.cat([g.reshape(-1).float() for g in grads])          # one buffer = one collective
        return loss.detach()

    # ---- the data-parallel form in stages: graph k = backward of block group k (+ the forward in graph 0) | all-reduce k (async) ----
    def _staged_forward(self):
        with self.staged:
            with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
                self._staged_loss = self.diffusion.training_losses(self.model, self.sz, self.st, dict(y=self.sy, y2=self.sy2, w=self.sw))["loss"].mean()
        return self._staged_loss.detach()

    def _staged_stage(self, k):
        self.staged.run_stage(k, self._staged_loss)
        ps = self.staged.params[k]
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in ps])

    def _staged_eager(self):
        import torch.distributed as dist
        self._staged_forward()
        self.flats = []
        works = []
        for k in range(self.staged.nstage):
            self.flats.append(self._staged_stage(k))
            works.append(dist.all_reduce(self.flats[k], op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
        for wk in works:
            wk.wait()
        self._update()

    def _all_reduce(self):
        import torch.distributed as dist
        dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.pg)

    def _update(self):
        with torch.no_grad():
            if self.staged is not None:
                groups = list(zip(self.staged.params, self.flats))
            else:
                groups = [(self._gp, self.flat)]
            found = None
            dst, src = [], []
            in_place = self._fused is not None and all(f.dtype == torch.float32 for _, f in groups)
            for ps, flat in groups:
                off = 0
                for p in ps:                                # the averaged gradients back into the tensors the optimizer reads
                    n = p.numel()
                    g = flat[off:off + n].view_as(p)
                    if in_place:                            # K14 reads the gradient through its table: the slice of the flat buffer IS the gradient
                        p.grad = g
                    elif p.grad is None:
                        p.grad = g.to(p.dtype).clone()
                    else:
                        dst.append(p.grad)
                        src.append(g)
                    off += n
                bad = (~torch.isfinite(flat).all()).float()         # after the all-reduce: the same on every rank
                found = bad if found is None else torch.maximum(found, bad)
            if dst:
                torch._foreach_copy_(dst, src)              # ONE multi-tensor launch (per-parameter copy_ was 459 launches per step)
        self._guarded_update(found)

    def step(self, z, t, y, y2, w):
        self.sz.copy_(z)
        self.st.copy_(t)
        self.sy.copy_(y)
        self.sy2.copy_(y2)
        self.sw.copy_(w)
        with torch.cuda.device(self.sz.device):
            if self.staged is not None:
                import torch.distributed as dist
                works = []
                for g, flat in zip(self.stage_graphs, self.flats):      # all-reduce k rides on RCCL's stream under graph k + 1
                    g.replay()
                    works.append(dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True))
                for wk in works:
                    wk.wait()
                self.graph2.replay()
            else:
                self.graph.replay()
                if self.split:
                    self._all_reduce()
                    self.graph2.replay()
        # a replayed optimizer updates A_log without bumping its version counter: drop the mixers' no-grad cache of -exp(A_log)
        for m in self._mixers:
            m.__dict__.pop("_A_cache", None)
        # ... and the 16-bit weight copies of step_prep are one optimizer step old for any EAGER forward that follows (the replayed
        # graph refreshes them itself at its top): mark them stale
        from . import step_prep
        step_prep.invalidate()
        return self.sloss
