"""K4x (conv backward + x_proj input gradient, dx merged over the directions): the slab form (running sum in LDS) against the
whole-sample form (running sum in HBM) at the bench shape -- run on the GPU box.  DM_K4X_SLAB picks the form per launch."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402

dev = torch.device("cuda", 0)
B, L, Dm, P = int(os.environ.get("B", 512)), int(os.environ.get("L", 196)), 1024, 64
dt = torch.bfloat16
g = torch.Generator(device="cpu").manual_seed(1)
xz = torch.randn(B, L, 2 * Dm, generator=g).to(dt).to(dev)
w, b = torch.randn(Dm, 4, generator=g).to(dev), torch.randn(Dm, generator=g).to(dev)
wxt = (torch.randn(Dm, P, generator=g) * 0.05).to(dt).to(dev)
idx = torch.stack([torch.arange(L), torch.randperm(L, generator=g), torch.randperm(L, generator=g)]).int().to(dev)
du = torch.randn(3 * B, L, Dm, generator=g).to(dt).to(dev)
dxd = torch.randn(3 * B * L, P, generator=g).to(dt).to(dev)
dxz = torch.zeros(B, L, 2 * Dm, dtype=dt, device=dev)


def run(slab):
    os.environ["DM_K4X_SLAB"] = slab
    return hip_ops.gather_conv1d_xproj_bwd(xz[..., :Dm], w, b, du, dxd, wxt, row_index=idx, ndir=3, merged_out=dxz[..., :Dm])


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


r0 = [t.clone() for t in run("0")]
r1 = [t.clone() for t in run("1")]
torch.cuda.synchronize()
for name, x0, x1 in zip(("dx", "dw", "db"), r0, r1):
    d = (x0.float() - x1.float()).abs()
    print(f"  {name}: max abs diff {float(d.max()):.4g}  mismatching {int((d > 0).sum())} of {d.numel()}  (scale {float(x0.float().abs().max()):.3g})")
t0, t1 = timeit(lambda: run("0")), timeit(lambda: run("1"))
alg = (3 * 2 + 1) * B * L * Dm * 2 + 3 * B * L * P * 2
print(json.dumps(dict(B=B, L=L, whole_sample_us=round(t0, 1), slab_us=round(t1, 1), note="host-timed, includes the column sum of dw|db",
                      slab_alg_GBps=round(alg / t1 / 1e3, 1))))
