import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops
dev = torch.device("cuda", 0)
for B in (8, 64, 256):
    y = torch.randn(3, B, 196, 1024, device=dev).bfloat16()
    w = torch.randn(1024, device=dev)
    out, rstd = hip_ops.rmsnorm_merge_fwd(y, w, 1e-5)
    dout = torch.randn_like(out)
    def f(): return hip_ops.rmsnorm_merge_bwd(y, w, 1e-5, rstd, dout)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    nb = 7 * B * 196 * 1024 * 2
    print(f"B {B}: rmsnorm_merge_bwd {us:.1f} us  {nb/us/1e3:.0f} GB/s")
