"""Gradient error of the Mamba-2 mixer against fp64 autograd through the oracle: matrix-pipe SSD pair (K6 / K6b) vs the A-shared
scan pair (K1 / K2), bf16 autocast -- run on the GPU box.  What the 16-bit rounding of the score / gradient tiles costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402
from diffma_amd.mamba2 import Mamba2  # noqa: E402
from diffma_amd.tools import spiral  # noqa: E402
from oracle.mamba2_ref import mamba2_spiral_forward_ref  # noqa: E402

dev = torch.device("cuda", 0)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
n, d_model = 14, 256
torch.manual_seed(0)
orders, inverses = spiral(n)
lists = (orders[2], orders[3], inverses[2], inverses[3])
mix = Mamba2(d_model=d_model, d_state=16, d_conv=4, expand=2, token_list=lists[0], token_list_reversal=lists[1], origina_list=lists[2],
             origina_list_reversal=lists[3]).to(dev)
x = torch.randn(4, n * n, d_model, device=dev, requires_grad=True)
dy = torch.randn(4, n * n, d_model, device=dev)
params = {k: v.detach().cpu().double().requires_grad_(True) for k, v in mix.state_dict().items()}
x64 = x.detach().cpu().double().requires_grad_(True)
yr = mamba2_spiral_forward_ref(x64, params, lists, headdim=64, dtype=torch.float64)
(yr * dy.cpu().double()).sum().backward()
for name, flag in (("matrix pipe (K6/K6b)", True), ("A-shared scans (K1/K2)", False)):
    hip_ops.SSD_MFMA = flag
    x.grad = None
    mix.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mix(x, "spiral")
    (y.float() * dy).sum().backward()
    errs = {k: rel(p.grad.cpu(), params[k].grad) for k, p in mix.named_parameters()}
    print(f"{name:24s} out {rel(y.detach().float().cpu(), yr.detach()):.2e}  dx {rel(x.grad.cpu(), x64.grad):.2e}  " +
          "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
