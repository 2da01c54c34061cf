import torch, sys, time
sys.path.insert(0, '.')
from diffma_amd import hip_ops
dev = torch.device('cuda', 0)
B, L, D = 256, 196, 1024
x = torch.randn(B, L, 2 * D, device=dev).to(torch.bfloat16)
w = torch.randn(D, 4, device=dev); b = torch.randn(D, device=dev)
idx = torch.stack([torch.arange(L), torch.randperm(L), torch.randperm(L)]).to(torch.int32).to(dev)
dout = torch.randn(3 * B, L, D, device=dev).to(torch.bfloat16)
f = lambda: hip_ops.gather_conv1d_bwd(x[..., :D], w, b, dout, row_index=idx, ndir=3, silu=True)
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): f()
torch.cuda.synchronize(); print("conv_bwd us", (time.perf_counter() - t0) / 20 * 1e6)
