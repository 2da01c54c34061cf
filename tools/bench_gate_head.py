"""dm_gate_head_fwd / bwd at the DiffMa-L/2 block shape (run on the GPU box): python tools/bench_gate_head.py [batch ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops  # noqa: E402
dev = torch.device("cuda", 0)


def timeit(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or [8, 512]:
    M, C = B * 196, 512
    h = torch.randn(M, C, device=dev).bfloat16()
    b1, w2, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * C ** -0.5, torch.zeros(1, device=dev)
    a = hip_ops.gate_head_fwd(h, b1, w2, b2)
    da = torch.randn(M, 1, device=dev).bfloat16()
    us_f = timeit(lambda: hip_ops.gate_head_fwd(h, b1, w2, b2))
    for nb in [int(v) for v in os.environ.get("GH_BLOCKS", "1024").split(",")]:
        hip_ops.GATE_HEAD_BWD_BLOCKS = nb
        us_b = timeit(lambda: hip_ops.gate_head_bwd(da, a, h, b1, w2))
        print(f"batch {B}: gate_head fwd {us_f:.1f} us ({M * C * 2 / us_f / 1e3:.0f} GB/s) | bwd + colsum nblk {nb}: {us_b:.1f} us ({2 * M * C * 2 / us_b / 1e3:.0f} GB/s)", flush=True)
