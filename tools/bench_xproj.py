"""Fused conv + x_proj (K3x) against the unfused pair at the bench shape -- run on the GPU box."""
import os, sys, json
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops, gemm_tuning  # noqa: E402

gemm_tuning.enable_tuned_gemms(tune_missing=False)
dev = torch.device("cuda", 0)
B, L, Dm, P = int(os.environ.get("B", 512)), 196, 1024, 64
dt = torch.bfloat16
xz = torch.randn(B, L, 2 * Dm, device=dev).to(dt)
w, b = torch.randn(Dm, 4, device=dev), torch.randn(Dm, device=dev)
wx = (torch.randn(P, Dm, device=dev) * 0.05).to(dt)
idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).int().to(dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def unfused():
    xc = hip_ops.gather_conv1d_fwd(xz[..., :Dm], w, b, row_index=idx, ndir=3)
    return xc, F.linear(xc.view(-1, Dm), wx)


def fused():
    return hip_ops.gather_conv1d_xproj_fwd(xz[..., :Dm], w, b, wx, row_index=idx, ndir=3)


a, c = unfused(), fused()
print("x~ equal:", torch.equal(a[0], c[0]), " x_dbl max abs diff:", float((a[1].float() - c[1].float()).abs().max()))
if not torch.equal(a[0], c[0]):
    d = (a[0].float() - c[0].float())
    bad = d.abs() > 0
    print("  mismatches:", int(bad.sum()), "of", d.numel(), " max abs:", float(d.abs().max()))
    nz = bad.nonzero()
    print("  first few (seq, l, ch):", nz[:8].tolist())
    print("  mismatching l histogram:", torch.bincount(nz[:, 1], minlength=L)[:40].tolist())
    print("  mismatching ch%8 histogram:", torch.bincount(nz[:, 2] % 8, minlength=8).tolist())
    i = nz[0].tolist()
    print("  values:", float(a[0][i[0], i[1], i[2]]), float(c[0][i[0], i[1], i[2]]))
t_conv = timeit(lambda: hip_ops.gather_conv1d_fwd(xz[..., :Dm], w, b, row_index=idx, ndir=3))
t_un, t_fu = timeit(unfused), timeit(fused)
nb = 2 * 3 * B * L * Dm * 2 + 3 * B * L * P * 2
print(json.dumps(dict(B=B, conv_only_us=round(t_conv, 1), unfused_pair_us=round(t_un, 1), fused_us=round(t_fu, 1),
                      fused_GBps=round(nb / t_fu / 1e3, 1))))

# ---- backward: in-place addmm + conv_bwd against the fused K4x --------------------------------------------------------
S = 3 * B
du = torch.randn(S, L, Dm, device=dev).to(dt)
dxd = torch.randn(S * L, P, device=dev).to(dt)
wxt = wx.t().contiguous()


def unfused_b():
    dxc = du.clone().view(-1, Dm).addmm_(dxd, wx).view(S, L, Dm)
    return hip_ops.gather_conv1d_bwd(xz[..., :Dm], w, b, dxc, row_index=idx, ndir=3)


def unfused_b_noclone():      # what the model does: addmm_ in place on du (du is consumed)
    scratch = du                # timing only: the values drift, the traffic is the same
    dxc = scratch.view(-1, Dm).addmm_(dxd, wx, beta=1.0, alpha=0.0).view(S, L, Dm)
    return hip_ops.gather_conv1d_bwd(xz[..., :Dm], w, b, dxc, row_index=idx, ndir=3)


def fused_b():
    return hip_ops.gather_conv1d_xproj_bwd(xz[..., :Dm], w, b, du, dxd, wxt, row_index=idx, ndir=3)


ra, rc = unfused_b(), fused_b()
for name, x0, x1 in zip(("dx", "dw", "db"), ra, rc):
    d = (x0.float() - x1.float()).abs()
    print(f"  bwd {name}: max abs diff {float(d.max()):.4g} (scale {float(x0.float().abs().max()):.3g})")
t_cb = timeit(lambda: hip_ops.gather_conv1d_bwd(xz[..., :Dm], w, b, du, row_index=idx, ndir=3))
t_ub, t_fb = timeit(unfused_b_noclone), timeit(fused_b)
print(json.dumps(dict(B=B, conv_bwd_only_us=round(t_cb, 1), unfused_bwd_pair_us=round(t_ub, 1), fused_bwd_us=round(t_fb, 1))))
