"""Repeat ONE slab-form K4x launch and look at the partial rows it writes (prefilled with a marker) -- run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops, _lib  # noqa: E402
from diffma_amd._lib import dm_conv_xproj_bwd_args, DM_FLAG_SILU, DM_FLAG_DX_MERGED  # noqa: E402

dev = torch.device("cuda", 0)
B, L, Dm, P, ND, W = 2, 196, int(os.environ.get("DM", 1024)), 64, 3, 4
dt = torch.bfloat16
os.environ["DM_K4X_SLAB"] = os.environ.get("SLAB", "1")
g = torch.Generator().manual_seed(3)
xz = torch.zeros(B, L, 2 * Dm, dtype=dt, device=dev)
xz[..., :Dm] = ((torch.arange(L).view(1, L, 1) + 256 * torch.arange(B).view(B, 1, 1)).float() / 256).to(dt).to(dev)
x = xz[..., :Dm]
w = torch.zeros(Dm, 4, device=dev)
b = torch.full((Dm,), 40.0, device=dev)
wxt = torch.zeros(Dm, P, dtype=dt, device=dev)
idx = torch.stack([torch.arange(L)] + [torch.randperm(L, generator=g) for _ in range(ND - 1)]).int().to(dev)
dxd = torch.zeros(ND * B * L, P, dtype=dt, device=dev)
du = torch.ones(ND * B, L, Dm, dtype=dt, device=dev)          # every row counts 1: db partial of a sample = 3 L exactly
lib = _lib.load()
want_db = float(ND * L)
fails = 0
for it in range(int(os.environ.get("ITERS", 2000))):
    part = torch.full((B, Dm * (W + 1)), 777.0, device=dev)
    dxz = torch.zeros(B, L, 2 * Dm, dtype=dt, device=dev)
    dx = dxz[..., :Dm]
    a = dm_conv_xproj_bwd_args()
    a.part_ss = Dm * (W + 1)
    a.batch, a.dim, a.seqlen, a.width, a.ndir = B, Dm, L, W, ND
    a.io_dtype, a.w_dtype = hip_ops.dtype_code(x), hip_ops.dtype_code(w)
    a.flags = DM_FLAG_SILU | DM_FLAG_DX_MERGED
    a.nproj = P
    a.x, a.weight, a.bias, a.row_index = x.data_ptr(), w.data_ptr(), b.data_ptr(), idx.data_ptr()
    a.du, a.dxdbl, a.wxt = du.data_ptr(), dxd.data_ptr(), wxt.data_ptr()
    a.dx, a.dw_partial, a.db_partial = dx.data_ptr(), part.data_ptr(), part[:, Dm * W:].data_ptr()
    a.x_sb, a.x_sl, a.x_sd = x.stride()
    a.du_ss, a.du_sl, a.du_sd = du.stride()
    a.dx_ss, a.dx_sl, a.dx_sd = dx.stride()
    a.xd_sr = dxd.stride(0)
    _lib.call("dm_gather_conv1d_xproj_bwd", a, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    db = part[:, Dm * W:]
    dwp = part[:, :Dm * W].view(B, Dm, W)
    if it == 0:
        ref_dw = dwp.clone()
    bad_db = (db != want_db)
    bad_dw = (dwp != ref_dw)
    bad_dx = (dx.float() != 3.0 * 0 + dx.float()[0, 0, 0])
    if bad_db.any() or bad_dw.any():
        fails += 1
        if fails <= 12:
            print("iter", it, "db bad at", bad_db.nonzero()[:6].tolist(), db[bad_db][:6].tolist(),
                  "| dw bad at", bad_dw.nonzero()[:6].tolist(), dwp[bad_dw][:6].tolist(), "ref", ref_dw[bad_dw][:6].tolist())
print("failing launches:", fails)
