"""dm_dtproj_softplus_fwd at the DiffMa-L/2 mixer shape against F.linear (run on the GPU box).
   DM_DTP_VARIANT=k python tools/bench_dtproj.py [nseq ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops, gemm_tuning  # noqa: E402,F401
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


for S in [int(a) for a in sys.argv[1:]] or [24, 192, 768, 1536]:
    M, Dm, R, P = S * 196, 1024, 32, 64
    xdbl = torch.randn(M, P, device=dev).bfloat16()
    w = (torch.randn(Dm, R, device=dev) * R ** -0.5).bfloat16()
    b = torch.randn(Dm, device=dev)
    us = timeit(lambda: hip_ops.dtproj_softplus_fwd(xdbl, w, b))
    us_lib = timeit(lambda: torch.nn.functional.linear(xdbl[:, :R], w))
    nb = M * (Dm + R) * 2
    print(f"variant {os.environ.get('DM_DTP_VARIANT', '0')} nseq {S}: dtproj_softplus {us:.1f} us {nb / us / 1e3:.0f} GB/s | F.linear {us_lib:.1f} us", flush=True)

# backward: dm_dtproj_bwd (one read of ddelta) against the two library products it replaces
from diffma_amd.selective_scan_interface import _tn_splitk  # noqa: E402
for S in [int(a) for a in sys.argv[1:]] or [24, 192, 768, 1536]:
    M, Dm, R, P = S * 196, 1024, 32, 64
    if M % 32:
        continue
    dd = torch.randn(M, Dm, device=dev).bfloat16()
    xdbl = torch.randn(M, P, device=dev).bfloat16()
    dxd = torch.empty(M, P, device=dev).bfloat16()
    w = (torch.randn(Dm, R, device=dev) * R ** -0.5).bfloat16()
    for nb in [int(v) for v in os.environ.get("DTB_BLOCKS", "256").split(",")]:
        hip_ops.DTPROJ_BWD_BLOCKS = nb
        us = timeit(lambda: hip_ops.dtproj_bwd(dd, xdbl, w, dxd))
        print(f"nseq {S}: dtproj_bwd (+ colsum) nblk {nb}: {us:.1f} us  {M * Dm * 2 / us / 1e3:.0f} GB/s", flush=True)

    def lib():
        dxd[:, :R] = torch.mm(dd, w)
        return _tn_splitk(dd, xdbl[:, :R])
    print(f"nseq {S}: mm + copy + split-K bmm + sum: {timeit(lib):.1f} us", flush=True)
