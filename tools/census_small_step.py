"""Where do the copy / fill launches of a small-batch DiffMa-L/2 training step come from?  (~315 copies + ~316 fills of
~1 100 graph nodes per step.)  One eager step under torch.profiler with Python stacks; every aten::copy_ / fill_ / zero_ that launched a
device kernel is attributed to the innermost frame inside this repository.  Run on the GPU box:
    python tools/census_small_step.py [--batch 1] > gpurun_out/census_b1.txt"""
import argparse, collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rerandomize_zero_init, synthetic_batch, update_ema  # noqa: E402
from diffma_amd.diffusion import create_diffusion  # noqa: E402
from diffma_amd.model import DiffMa_models  # noqa: E402
import copy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DiffMa_models["DiffMa-L/2"](input_size=28, dt_rank=16, d_state=16)
rerandomize_zero_init(model, 1)
model = model.to(dev).train()
ema = copy.deepcopy(model).requires_grad_(False)
diffusion = create_diffusion("")
B = a.batch
gen = torch.Generator(device=dev).manual_seed(0)
batch = synthetic_batch(B, model.x_embedder.num_patches, dev, gen)
kw = dict(y=batch["y"], y2=batch["y2"], w=batch["w"])
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0, fused=True)


def step():
    t = torch.randint(0, diffusion.num_timesteps, (B,), device=dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = diffusion.training_losses(model, batch["z"], t, kw)["loss"].mean()
    loss.backward()
    opt.step()
    update_ema(ema, model)
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WATCH = ("copy_", "fill_", "zero_", "zeros", "zeros_like", "clone", "_to_copy", "new_zeros", "full", "ones_like", "empty_like", "add", "add_", "sum", "cat", "mul", "index", "neg", "exp")
by = collections.Counter()


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            t = next((x for x in args if isinstance(x, torch.Tensor)), None)
            if t is None or t.is_cuda or name in ("zeros", "full"):
                where = "(no repository frame: autograd engine / torch internals)"
                for fr in reversed(traceback.extract_stack()):
                    if fr.filename.startswith(root) and "/tools/census" not in fr.filename:
                        where = f"{fr.filename.replace(root + '/', '')}:{fr.lineno} {fr.name}"
                        break
                shape = "" if t is None else f"{tuple(t.shape)} {str(t.dtype).replace('torch.', '')}"
                by[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
torch.cuda.synchronize()
print(f"batch {B}: aten copy / fill / zero / clone / cast calls on device tensors in one eager DiffMa-L/2 training step, by innermost repository frame\n")
tot = collections.Counter()
agg = collections.Counter()
for (name, where, shape), n in by.items():
    tot[name] += n
    agg[(name, where)] += n
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    shapes = sorted(((sh, c) for (nm, wh, sh), c in by.items() if nm == name and wh == where), key=lambda kv: -kv[1])[:3]
    print(f"{n:5d}  {name:11s} {where:95s} {shapes}")
print("\ntotals:", dict(tot))
