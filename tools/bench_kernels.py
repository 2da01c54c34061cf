"""Micro-benchmark of the HIP kernels (GB/s of ALGORITHMIC bytes) -- run on the GPU box.

  python tools/bench_kernels.py [--batch 64] [--dtype fp32|bf16] [--iters 50]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402


def timeit(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 8, 64, 192, 384])
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--L", type=int, default=196)
    ap.add_argument("--D", type=int, default=1024)
    ap.add_argument("--only", default="", help="comma list of kernels: scan_fwd,scan_bwd,conv,merge,copy")
    args = ap.parse_args()
    dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    s = 4 if dt == torch.float32 else 2
    dev = torch.device("cuda", 0)
    L, Dm, N = args.L, args.D, 16
    res = []
    only = set(filter(None, args.only.split(",")))
    want = lambda k: not only or k in only
    for S in args.batch:
        u = torch.randn(S, L, Dm, device=dev).to(dt)
        delta = (torch.randn(S, L, Dm, device=dev) * 0.5).to(dt)
        z = torch.randn(S, L, Dm, device=dev).to(dt)
        A = -(torch.rand(Dm, N, device=dev) * 4 + 0.2)
        Bm = torch.randn(S, L, N, device=dev).to(dt)
        Cm = torch.randn(S, L, N, device=dev).to(dt)
        Dp = torch.randn(Dm, device=dev)
        bias = torch.randn(Dm, device=dev) * 0.5
        out = torch.empty_like(u)
        if want("scan_fwd"):
            t = timeit(lambda: hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, z, bias, True, out=out), args.iters)
            nbytes = 4 * S * Dm * L * s + 2 * S * N * L * s + 4 * Dm * N + 8 * Dm
            r = dict(kernel="scan_fwd", S=S, dtype=args.dtype, us=t * 1e6, GBps=nbytes / t / 1e9,
                     frac_8TBps=nbytes / t / 8e12, Gelem_s=S * Dm * L / t / 1e9)
            print(json.dumps(r), flush=True)
            res.append(r)
        if want("scan_bwd"):
            K = hip_ops.SCAN_CKPT_EVERY
            ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, dt, dev)
            hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, z, bias, True, out=out, ckpt=ckpt, ckpt_every=K)
            t = timeit(lambda: hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, z, bias, True, out=out, ckpt=ckpt, ckpt_every=K), args.iters)
            print(json.dumps(dict(kernel="scan_fwd_ckpt", S=S, dtype=args.dtype, us=t * 1e6, Gelem_s=S * Dm * L / t / 1e9)), flush=True)
            dout = torch.randn(S, L, Dm, device=dev).to(dt)
            t = timeit(lambda: hip_ops.scan_bwd(u, delta, A, Bm, Cm, Dp, z, bias, dout, ckpt, True, ckpt_every=K), args.iters)
            nbytes = 7 * S * Dm * L * s + 2 * S * N * L * s
            print(json.dumps(dict(kernel="scan_bwd(+partial sums)", S=S, dtype=args.dtype, us=t * 1e6, GBps=nbytes / t / 1e9,
                                  Gelem_s=S * Dm * L / t / 1e9)), flush=True)
        if want("scan_idx") and S % 3 == 0:
            # the model's call: 3 directions x batch in one launch, z / dout shared and read through the row-index table
            Bd = S // 3
            K = hip_ops.SCAN_CKPT_EVERY
            idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).to(torch.int32).to(dev)
            zb, doutb = z[:Bd].contiguous(), torch.randn(Bd, L, Dm, device=dev).to(dt)
            ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, dt, dev)
            kw = dict(z_row_index=idx, out_row_index=idx, batch_per_dir=Bd)
            t = timeit(lambda: hip_ops.scan_fwd(u, delta, A, Bm, Cm, Dp, zb, bias, True, out=out, ckpt=ckpt, ckpt_every=K, **kw), args.iters)
            print(json.dumps(dict(kernel="scan_fwd_idx_ckpt", S=S, dtype=args.dtype, us=t * 1e6)), flush=True)
            t = timeit(lambda: hip_ops.scan_bwd(u, delta, A, Bm, Cm, Dp, zb, bias, doutb, ckpt, True, ckpt_every=K, **kw), args.iters)
            print(json.dumps(dict(kernel="scan_bwd_idx(+partial sums)", S=S, dtype=args.dtype, us=t * 1e6)), flush=True)
        if want("scan_hoist") and S % 3 == 0:
            # the mixer's call since round 3: gate and softplus hoisted -- no z, delta already activated, pre-gated gradient
            Bd = S // 3
            K = hip_ops.SCAN_CKPT_EVERY
            idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).to(torch.int32).to(dev)
            act = torch.nn.functional.softplus(delta.float() + bias).to(dt)
            doutb = torch.randn(Bd, L, Dm, device=dev).to(dt)
            ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Dm, dt, dev)
            kw = dict(z_row_index=idx, out_row_index=idx, batch_per_dir=Bd, delta_activated=True)
            t = timeit(lambda: hip_ops.scan_fwd(u, act, A, Bm, Cm, Dp, None, bias, True, out=out, ckpt=ckpt, ckpt_every=K, **kw), args.iters)
            print(json.dumps(dict(kernel="scan_fwd_hoisted_ckpt", S=S, dtype=args.dtype, us=t * 1e6)), flush=True)
            t = timeit(lambda: hip_ops.scan_bwd(u, act, A, Bm, Cm, Dp, None, bias, doutb, ckpt, True, ckpt_every=K, **kw), args.iters)
            print(json.dumps(dict(kernel="scan_bwd_hoisted(+partial sums)", S=S, dtype=args.dtype, us=t * 1e6)), flush=True)
            tm = hip_ops.KernelTimer()                          # the kernel alone (events around the C-ABI launch)
            prev = hip_ops.set_timer(tm)
            for _ in range(args.iters):
                hip_ops.scan_bwd(u, act, A, Bm, Cm, Dp, None, bias, doutb, ckpt, True, ckpt_every=K, **kw)
            hip_ops.set_timer(prev)
            print(json.dumps(dict(kernel="scan_bwd_hoisted_kernel_only", S=S, dtype=args.dtype, us=tm.summary()["dm_selective_scan_bwd"]["avg_us"])), flush=True)
        if not (want("conv") or want("merge") or want("copy")):
            continue
        # conv (3 directions) : reads x once per direction, writes 3 outputs
        xz = torch.randn(S, L, 2 * Dm, device=dev).to(dt)
        w = torch.randn(Dm, 4, device=dev)
        b = torch.randn(Dm, device=dev)
        idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).int().to(dev)
        o3 = torch.empty(3 * S, L, Dm, device=dev, dtype=dt)
        t = timeit(lambda: hip_ops.gather_conv1d_fwd(xz[..., :Dm], w, b, row_index=idx, ndir=3, out=o3), args.iters)
        nbytes = 3 * 2 * S * Dm * L * s
        r = dict(kernel="gather_conv_fwd_x3", S=S, dtype=args.dtype, us=t * 1e6, GBps=nbytes / t / 1e9)
        print(json.dumps(r), flush=True)
        slabs = o3.view(3, S, L, Dm)
        om = torch.empty(S, L, Dm, device=dev, dtype=dt)
        t = timeit(lambda: hip_ops.token_merge(slabs, out=om), args.iters)
        nbytes = 4 * S * Dm * L * s
        r = dict(kernel="token_merge_x3", S=S, dtype=args.dtype, us=t * 1e6, GBps=nbytes / t / 1e9)
        print(json.dumps(r), flush=True)
        # reference point: device copy of the same footprint
        t = timeit(lambda: out.copy_(u), args.iters)
        print(json.dumps(dict(kernel="torch_copy", S=S, us=t * 1e6, GBps=2 * S * Dm * L * s / t / 1e9)), flush=True)


if __name__ == "__main__":
    main()
