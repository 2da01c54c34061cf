"""hipGraph capture of the denoiser step for sampling.

A DiffMa forward at batch 1-8 is ~400 kernel launches of a few microseconds each: launch-bound, exactly the
regime SURVEY.md 7.3 describes.  Shapes are static across all 250 (or 50) sampling steps, every C-ABI launch is
asynchronous and allocation-free, and the diffusion tables live on the device, so the whole
`model(x, t, y, y2, w)` call is captured ONCE into a HIP graph (torch.cuda.CUDAGraph drives hipGraph on ROCm)
and replayed per step with the inputs copied into static buffers.
"""
from __future__ import annotations

import torch


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
            with torch.no_grad(), torch.cuda.graph(self.graph):
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
    Philox offsets advance per replay).  Construction leaves weights, EMA and optimizer state as it found them.  Single
    process only: under DDP the bucketed all-reduce keeps the eager step.

    step(z, t, y, y2, w) -> loss (a 0-d device tensor that is overwritten by the next replay).
    """

    def __init__(self, model, ema, optimizer, diffusion, z, t, y, y2, w, autocast_dtype=None, ema_decay=0.9999, warmup=3):
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
        with torch.cuda.device(z.device):
            side = torch.cuda.Stream(device=z.device)
            side.wait_stream(torch.cuda.current_stream(z.device))
            with torch.cuda.stream(side):
                for _ in range(warmup):                   # lazy inits, GEMM solution lookups, optimizer state allocation
                    self._step()
            torch.cuda.current_stream(z.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
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
        self.opt.step()
        if self.ema is not None:
            with torch.no_grad():
                torch._foreach_mul_(self._ep, self.decay)
                torch._foreach_add_(self._ep, self._mp, alpha=1 - self.decay)
        self.opt.zero_grad(set_to_none=True)
        return loss.detach()

    def step(self, z, t, y, y2, w):
        self.sz.copy_(z)
        self.st.copy_(t)
        self.sy.copy_(y)
        self.sy2.copy_(y2)
        self.sw.copy_(w)
        with torch.cuda.device(self.sz.device):
            self.graph.replay()
        # a replayed optimizer updates A_log without bumping its version counter: drop the mixers' no-grad cache of -exp(A_log)
        for m in self._mixers:
            m.__dict__.pop("_A_cache", None)
        return self.sloss
