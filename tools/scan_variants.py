"""Forced-variant A/B of the scan kernels at the bench shape: the sequential-in-lane kernels (K1 / K2) against the
time-parallel two-pass kernels (K1c / K2c, the north star's "wave-parallel" form) in the mixer's call pattern -- bf16 I/O, three
directions through row-index tables, no z, delta already activated, checkpoints for the backward.  Kernel-only times (events around
the C-ABI launch).

  python tools/scan_variants.py [--nseq 1536 768 96 24] [--iters 10]  > profiles/r04_scan_variants.txt
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402


def kernel_us(fn, name, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    tm = hip_ops.KernelTimer()
    prev = hip_ops.set_timer(tm)
    for _ in range(iters):
        fn()
    hip_ops.set_timer(prev)
    torch.cuda.synchronize()
    return tm.summary()[name]["avg_us"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nseq", type=int, nargs="+", default=[1536, 768, 96, 24])
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L, Dm, N, dt = 196, 1024, 16, torch.bfloat16
    for S in a.nseq:
        Bd = S // 3
        g = torch.Generator(device=dev).manual_seed(S)
        mk = lambda *s: torch.randn(*s, device=dev, generator=g)
        u, Bm, Cm = mk(S, L, Dm).to(dt), mk(S, L, N).to(dt), mk(S, L, N).to(dt)
        bias = mk(Dm) * 0.5
        act = torch.nn.functional.softplus(mk(S, L, Dm) * 0.5 + bias).to(dt)
        A, Dp = -(torch.rand(Dm, N, device=dev, generator=g) * 4 + 0.2), mk(Dm)
        dout = mk(Bd, L, Dm).to(dt)
        idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).to(torch.int32).to(dev)
        out = torch.empty_like(u)
        kw = dict(z_row_index=idx, out_row_index=idx, batch_per_dir=Bd, delta_activated=True)
        row = dict(nseq=S, L=L, D=Dm, N=N, dtype="bf16")
        for variant in ("sequential", "chunked"):
            ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, dt, dev)
            f = lambda: hip_ops.scan_fwd(u, act, A, Bm, Cm, Dp, None, bias, True, out=out, ckpt=ckpt, variant=variant, **kw)
            f()
            b = lambda: hip_ops.scan_bwd(u, act, A, Bm, Cm, Dp, None, bias, dout, ckpt, True, variant=variant, **kw)
            row[f"fwd_ckpt_{variant}_us"] = round(kernel_us(f, "dm_selective_scan_fwd", a.iters), 1)
            row[f"bwd_{variant}_us"] = round(kernel_us(b, "dm_selective_scan_bwd", a.iters), 1)
            f2 = lambda: hip_ops.scan_fwd(u, act, A, Bm, Cm, Dp, None, bias, True, out=out, variant=variant, **kw)
            try:
                row[f"fwd_nockpt_{variant}_us"] = round(kernel_us(f2, "dm_selective_scan_fwd", a.iters), 1)
            except Exception as e:                       # the chunk-parallel forward writes no checkpoints: it is the no-grad kernel
                row[f"fwd_nockpt_{variant}_us"] = f"n/a ({type(e).__name__})"
            del ckpt
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
