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
                 process_group=None, split=None):
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
            o_snap = {id(v): v.detach().clone() for st in optimizer.state.values() for v in st.values() if torch.is_tensor(v)}
        import torch.distributed as dist
        self.pg = process_group
        nranks = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.split = bool(split) if split is not None else nranks > 1
        if self.split and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("the two-graph (data-parallel) step needs an initialised process group")
        self._gp = [p for p in model.parameters() if p.requires_grad]
        # Non-finite gradients (the reference's eager loop skips such a step, train.py:254-256): a replayed graph cannot branch on
        # the host, so the decision is taken ON THE DEVICE -- found_inf is computed from the (all-reduced, hence rank-identical)
        # gradients inside the graph and handed to the fused AdamW, which then leaves weights, moments and step count alone.
        # `skipped` counts such steps; the host reads it when it logs.  The EMA blend is gated by the same device flag (the reference
        # `continue`s before update_ema, train.py:254-264): ema <- model + d * (ema - model) with d = decay, or 1 on a dropped step.
        self._one = torch.ones((), device=z.device)
        self.skipped = torch.zeros((), device=z.device)
        with torch.cuda.device(z.device):
            side = torch.cuda.Stream(device=z.device)
            side.wait_stream(torch.cuda.current_stream(z.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):                   # lazy inits, GEMM solution lookups, optimizer state allocation
                    if self.split:
                        self._fwd_bwd()
                        self._all_reduce()
                        self._update()
                    else:
                        self._step()
            torch.cuda.current_stream(z.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            mode = _capture_mode(z.device)
            if self.split:
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

    def _step(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            loss = self.diffusion.training_losses(self.model, self.sz, self.st, dict(y=self.sy, y2=self.sy2, w=self.sw))["loss"].mean()
        loss.backward()
        found = torch.zeros((), device=self.sz.device)            # 0-dim like GradScaler's (the fused AdamW subtracts it from its 0-dim step counters)
        grads = [p.grad for p in self._gp if p.grad is not None]
        if grads:
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found, self._one)      # inv_scale 1: a pure check
        self._guarded_update(found)
        self.opt.zero_grad(set_to_none=True)
        return loss.detach()

    def _guarded_update(self, found):
        self.opt.grad_scale, self.opt.found_inf = None, found          # read by the fused AdamW: found_inf = 1 -> no update
        self.opt.step()
        with torch.no_grad():
            self.skipped += found
            if self.ema is not None:
                d = found * (1.0 - self.decay) + self.decay          # 0-dim device tensor: decay, or 1.0 when the step was dropped
                torch._foreach_sub_(self._ep, self._mp)
                torch._foreach_mul_(self._ep, d)
                torch._foreach_add_(self._ep, self._mp)

    # ---- the data-parallel form: graph 1 | all-reduce | graph 2 ----------------------------------------------------------
    def _fwd_bwd(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            loss = self.diffusion.training_losses(self.model, self.sz, self.st, dict(y=self.sy, y2=self.sy2, w=self.sw))["loss"].mean()
        loss.backward()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self._gp]
        self.flat = torch.cat([g.reshape(-1).float() for g in grads])          # one buffer = one collective
        return loss.detach()

    def _all_reduce(self):
        import torch.distributed as dist
        dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.pg)

    def _update(self):
        with torch.no_grad():
            off = 0
            for p in self._gp:                              # the averaged gradients back into the tensors the optimizer reads
                n = p.numel()
                g = self.flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.to(p.dtype).clone()
                else:
                    p.grad.copy_(g)
                off += n
            found = (~torch.isfinite(self.flat).all()).float()     # after the all-reduce: the same on every rank
        self._guarded_update(found)

    def step(self, z, t, y, y2, w):
        self.sz.copy_(z)
        self.st.copy_(t)
        self.sy.copy_(y)
        self.sy2.copy_(y2)
        self.sw.copy_(w)
        with torch.cuda.device(self.sz.device):
            self.graph.replay()
            if self.split:
                self._all_reduce()
                self.graph2.replay()
        # a replayed optimizer updates A_log without bumping its version counter: drop the mixers' no-grad cache of -exp(A_log)
        for m in self._mixers:
            m.__dict__.pop("_A_cache", None)
        return self.sloss
