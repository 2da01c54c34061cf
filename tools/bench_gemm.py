"""Skinny-GEMM shapes of the mixer (x_proj / dt_proj and their backward) on hipBLASLt vs rocBLAS."""
import json
import sys
import torch

def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda", 0)
M, Din, R, N = int(sys.argv[1]) if len(sys.argv) > 1 else 150528, 1024, 32, 16
dt = torch.bfloat16
xc = torch.randn(M, Din, device=dev, dtype=dt)
Wx = torch.randn(R + 2 * N, Din, device=dev, dtype=dt)
Wdt = torch.randn(Din, R, device=dev, dtype=dt)
xdbl = torch.randn(M, R + 2 * N, device=dev, dtype=dt)
dd = torch.randn(M, Din, device=dev, dtype=dt)
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    res = {}
    res["x_proj fwd  (M,1024)@(1024,64)"] = timeit(lambda: torch.nn.functional.linear(xc, Wx))
    res["dt_proj fwd (M,32)@(32,1024)"] = timeit(lambda: torch.nn.functional.linear(xdbl[:, :R], Wdt))
    res["dt_proj fwd contiguous input"] = timeit(lambda: torch.nn.functional.linear(xdbl[:, :R].contiguous(), Wdt))
    res["d(xdbl)=dd@Wdt (M,1024)@(1024,32)"] = timeit(lambda: dd @ Wdt)
    res["dWdt=dd^T@xdbl (1024,M)@(M,32)"] = timeit(lambda: dd.t() @ xdbl[:, :R])
    res["dWx=xdbl^T@xc (64,M)@(M,1024)"] = timeit(lambda: xdbl.t() @ xc)
    res["dxc=addmm(du, xdbl, Wx) (M,64)@(64,1024)"] = timeit(lambda: torch.addmm(dd, xdbl, Wx))
    res["in_proj (M/3,512)@(512,2048)"] = timeit(lambda: torch.nn.functional.linear(xc[: M // 3, :512], Wx.new_empty(2048, 512).normal_()))
    print(lib, json.dumps({k: round(v, 1) for k, v in res.items()}))
    ideal = {"x_proj": M * Din * 2 / 5e12 * 1e6, "out (M,1024) bf16 write": M * Din * 2 / 5e12 * 1e6}
print("ideal us at 5 TB/s for one (M,1024) bf16 pass:", round(M * Din * 2 / 5e12 * 1e6, 1))

torch.backends.cuda.preferred_blas_library("cublaslt")
def splitk(a, b, C):      # a: (M, P), b: (M, Q) -> a^T @ b  (P, Q) with K = M split in C chunks
    Mloc = a.shape[0]
    pa = a.view(C, Mloc // C, a.shape[1]).transpose(1, 2)
    pb = b.view(C, Mloc // C, b.shape[1])
    return torch.bmm(pa, pb).float().sum(0)
xr = xdbl[:, :R].contiguous()
for C in (8, 16, 32, 64, 128, 256):
    r = {"C": C, "dWx splitK": round(timeit(lambda: splitk(xdbl, xc, C)), 1), "dWdt splitK": round(timeit(lambda: splitk(dd, xr, C)), 1)}
    print(json.dumps(r))
ref = (xdbl.float().t() @ xc.float())
got = splitk(xdbl, xc, 64)
print("dWx splitK rel err vs fp32:", float((got - ref).norm() / ref.norm()), " plain bf16 gemm rel err:", float(((xdbl.t() @ xc).float() - ref).norm() / ref.norm()))

# weight gradients of the wide projections (K = B*L = 50176 rows): plain GEMM vs split-K
Mp = M // 3
dY = torch.randn(Mp, 2048, device=dev, dtype=dt); Xp = torch.randn(Mp, 512, device=dev, dtype=dt)
dYo = torch.randn(Mp, 512, device=dev, dtype=dt); Xo = torch.randn(Mp, 1024, device=dev, dtype=dt)
print(json.dumps({"in_proj dW plain (2048,M)@(M,512)": round(timeit(lambda: dY.t() @ Xp), 1),
                  "out_proj dW plain (512,M)@(M,1024)": round(timeit(lambda: dYo.t() @ Xo), 1)}))
for C in (4, 8, 16, 32):
    print(json.dumps({"C": C, "in_proj dW splitK": round(timeit(lambda: splitk(dY, Xp, C)), 1),
                      "out_proj dW splitK": round(timeit(lambda: splitk(dYo, Xo, C)), 1)}))
