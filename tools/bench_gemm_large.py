"""K12 (csrc/gemm_large.hip) against the library at the projections' large-batch shapes -- run on the GPU box.
python tools/bench_gemm_large.py [--check]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402

dev = torch.device("cuda", 0)
M = int(os.environ.get("M", 100352))
SHAPES = [("in_proj fwd", M, 2048, 512), ("out_proj fwd", M, 512, 1024), ("in_proj dgrad", M, 512, 2048), ("out_proj dgrad", M, 1024, 512),
          ("square-ish", 8192, 8192, 4096)]
check = "--check" in sys.argv
if os.environ.get("TUNED", "0") == "1":        # the library as the bench uses it: TunableOp's best solution per shape
    from diffma_amd.gemm_tuning import enable_tuned_gemms
    enable_tuned_gemms()
ONLY = os.environ.get("ONLY", "")
NOLIB = os.environ.get("NOLIB", "0") == "1"


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, m, n, k in SHAPES:
    if ONLY and ONLY not in name:
        continue
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.rand(m, k, generator=g, device=dev) * 2 - 1).bfloat16()
    b = (torch.rand(n, k, generator=g, device=dev) * 2 - 1).bfloat16()
    fill = os.environ.get("FILL", "")        # operand data and the chip's power limit: "ones" / "zeros" / "small" (random integers -2..2)
    if fill == "ones":
        a.fill_(1.0); b.fill_(1.0)
    elif fill == "zeros":
        a.zero_(); b.zero_()
    elif fill == "small":
        a = torch.randint(-2, 3, (m, k), generator=g, device=dev).bfloat16(); b = torch.randint(-2, 3, (n, k), generator=g, device=dev).bfloat16()
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    ok = hip_ops.gemm_large_supported(a, b, out)
    t_lib = 1.0 if NOLIB else timeit(lambda: torch.nn.functional.linear(a, b))
    if not NOLIB and os.environ.get("TUNED", "0") == "1" and m > 50000:
        bt = b.t().contiguous()                                   # the library's own dgrad form: dy [M, K] @ W [K, N] (row-major W)
        t_nn = timeit(lambda: torch.mm(a, bt))
        print(f"   (library, NN form a @ W[K,N]: {t_nn:8.1f} us)", flush=True)
    fl = 2.0 * m * n * k
    line = f"{name:16s} M {m:7d} N {n:5d} K {k:5d} | library {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF"
    if ok:
        t_own = timeit(lambda: hip_ops.gemm_large(a, b, out))
        line += f" | K12 {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF  ({t_lib / t_own:.2f}x)"
        if check:
            ref = torch.nn.functional.linear(a, b).float()
            got = hip_ops.gemm_large(a, b, out).float()
            rel = float((got - ref).norm() / ref.norm())
            rows = torch.randint(0, m, (64,), device=dev)
            r64 = a[rows].double() @ b.double().t()
            rel64 = float((got[rows].double() - r64).norm() / r64.norm())
            line += f" | rel vs lib {rel:.2e}, vs fp64 rows {rel64:.2e}"
    print(line, flush=True)
