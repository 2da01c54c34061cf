"""Mamba-2 SSD forward on the matrix pipe (K6) against the A-shared scan at the DiffMa-XL/2 --use-mamba2 shape -- run on the GPU box."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402

dev = torch.device("cuda", 0)
L, H, P, N = 196, 16, 64, 16
Din = H * P
dt_ = torch.bfloat16


def timeit(fn, iters=int(os.environ.get("SSD_ITERS", "30"))):
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


for B in [int(a) for a in (sys.argv[1:] or ["8", "64", "256"])]:
    S = 3 * B
    xBC = torch.randn(S, L, Din + 2 * N, device=dev).to(dt_)
    dt_tok = (torch.randn(B, L, H, device=dev) * 0.7 - 1.0).to(dt_)
    z = torch.randn(B, L, Din, device=dev).to(dt_)
    A_h = -(torch.rand(H, device=dev) * 6 + 0.3)
    D_h, b_h = torch.randn(H, device=dev), torch.randn(H, device=dev) * 0.5
    idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).int().to(dev)
    x, Bm, Cm = xBC[..., :Din], xBC[..., Din:Din + N], xBC[..., Din + N:]
    out = torch.empty(S, L, Din, device=dev, dtype=dt_)

    def mfma():
        return hip_ops.ssd_fwd(x, Bm, Cm, dt_tok, z, A_h, D_h, b_h, z_row_index=idx, out_row_index=idx, batch_per_dir=B, out=out)

    A = A_h.repeat_interleave(P)[:, None].expand(Din, N).contiguous()
    Dp, bp = D_h.repeat_interleave(P), b_h.repeat_interleave(P)
    idx64 = idx.long()

    def scan():
        dtg = torch.stack([dt_tok[:, idx64[k]] for k in range(3)]).reshape(S, L, H, 1).expand(S, L, H, P).reshape(S, L, Din)
        return hip_ops.scan_fwd(x, dtg, A, Bm, Cm, Dp, z, bp, True, z_row_index=idx, out_row_index=idx, batch_per_dir=B, a_shared=True, out=out)

    # ---- backward: K6b against the A-shared scan backward (which also needs the forward's checkpoints and the expanded delta) ----
    dout = torch.randn(S, L, Din, device=dev).to(dt_)
    dxBC = torch.empty_like(xBC)

    def mfma_bwd():
        return hip_ops.ssd_bwd(x, Bm, Cm, dt_tok, z, dout, A_h, D_h, b_h, z_row_index=idx, out_row_index=idx, batch_per_dir=B, dx_out=dxBC[..., :Din])

    dtg = torch.stack([dt_tok[:, idx64[k]] for k in range(3)]).reshape(S, L, H, 1).expand(S, L, H, P).reshape(S, L, Din)
    ckpt = hip_ops.alloc_scan_ckpt(S, L, N, Din, dt_, dev)
    hip_ops.scan_fwd(x, dtg, A, Bm, Cm, Dp, z, bp, True, z_row_index=idx, out_row_index=idx, batch_per_dir=B, a_shared=True, out=out, ckpt=ckpt)

    def scan_bwd():
        return hip_ops.scan_bwd(x, dtg, A, Bm, Cm, Dp, z, bp, dout, ckpt, True, z_row_index=idx, out_row_index=idx, batch_per_dir=B, dout_per_seq=True,
                                du_out=dxBC[..., :Din], a_shared=True, dbc_out=dxBC[..., Din:])

    t_mb, t_sb = timeit(mfma_bwd), timeit(scan_bwd)
    a = mfma().float().clone()
    b = scan().float()
    nb = 3 * S * L * Din * 2
    t_m, t_s = timeit(mfma), timeit(scan)
    print(json.dumps(dict(batch=B, nseq=S, ssd_mfma_us=round(t_m, 1), a_shared_scan_us=round(t_s, 1), mfma_GBps=round(nb / t_m / 1e3, 1),
                          ssd_bwd_mfma_us=round(t_mb, 1), a_shared_scan_bwd_us=round(t_sb, 1),
                          max_abs_diff=float((a - b).abs().max()), scale=float(b.abs().max()))))
