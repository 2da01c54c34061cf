"""dX = dY @ W as an NN GEMM vs F.linear(dY, W^T contiguous) (NT) for the projection shapes at M = 512*196 rows, bf16."""
import sys, torch
sys.path.insert(0, ".")
def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)
if len(sys.argv) > 1 and sys.argv[1] == "tuned":
    from diffma_amd import gemm_tuning
    gemm_tuning.enable_tuned_gemms()
dev = torch.device("cuda", 0)
M = 512 * 196
for name, K, N in (("out_proj / MLP dX", 512, 1024), ("in_proj dX", 2048, 512), ("MLP.1 fwd-like", 1024, 512), ("x_proj-ish dX", 64, 1024), ("dt dX", 1024, 32)):
    dY = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(K, N, device=dev, dtype=torch.bfloat16)          # dX = dY @ W
    Wt = W.t().contiguous()
    print(f"{name:20s} K={K:5d} N={N:5d}  NN {timeit(lambda: dY @ W):7.1f} us   NT {timeit(lambda: torch.nn.functional.linear(dY, Wt)):7.1f} us")
