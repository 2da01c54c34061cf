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
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # lazy inits (hipBLASLt heuristics, allocator) happen outside the capture
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.sout = self._run()

    def _run(self):
        with torch.autocast("cuda", dtype=self.amp, enabled=self.amp is not None):
            return self.model(self.sx, self.st, y=self.sy, y2=self.sy2, w=self.sw)

    def __call__(self, x, t, y=None, y2=None, w=None, **kw):
        self.sx.copy_(x)
        self.st.copy_(t)
        cid = (y.data_ptr(), y2.data_ptr(), w.data_ptr())
        if cid != self._cond_id:
            self.sy.copy_(y)
            self.sy2.copy_(y2)
            self.sw.copy_(w)
            self._cond_id = cid
        self.graph.replay()
        return self.sout

    def parameters(self):
        return self.model.parameters()
