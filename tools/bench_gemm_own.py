"""dm_gemm against the vendor library at the paired-mixer path's shapes (run on the GPU box): two library GEMMs vs ONE paired launch.
   python tools/bench_gemm_own.py [B ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffma_amd import hip_ops
from diffma_amd.gemm_tuning import enable_tuned_gemms

enable_tuned_gemms()
dev = torch.device("cuda", 0)


def t_us(fn, reps=20):
    """DEVICE time of one call: `reps` calls captured in a hipGraph and replayed (eager timing of 10-us kernels measures the host)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def pair(fn):
    def run():
        with hip_ops.paired() as pr:
            fn(0)
            pr.second()
            fn(1)
    return run


for B in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 64]:
    M = B * 196
    mk = lambda *s: [torch.randn(*s, device=dev).bfloat16() for _ in (0, 1)]
    for name, K, N, rows in (("in_proj", 512, 2048, M), ("out_proj", 1024, 512, M), ("x_proj", 1024, 64, 3 * M)):
        x, W, dy = mk(rows, K), mk(N, K), mk(rows, N)
        y, dx = mk(rows, N), mk(rows, K)
        dW = [torch.empty(N, K, device=dev) for _ in (0, 1)]
        lib_f = t_us(lambda: [torch.mm(x[k], W[k].t(), out=y[k]) for k in (0, 1)])
        own_f = t_us(pair(lambda k: hip_ops.gemm(x[k], W[k], out=y[k])))
        lib_d = t_us(lambda: [torch.mm(dy[k], W[k], out=dx[k]) for k in (0, 1)])
        own_d = t_us(pair(lambda k: hip_ops.gemm(dy[k], W[k], True, False, out=dx[k])))
        lib_w = t_us(lambda: [torch.mm(dy[k].t(), x[k], out_dtype=torch.float32) for k in (0, 1)])
        own_w = t_us(pair(lambda k: hip_ops.gemm(dy[k], x[k], False, False, out=dW[k])))
        print(f"B={B:3d} {name:8s} rows={rows:6d}  fwd lib {lib_f:7.1f} own {own_f:7.1f} | dgrad lib {lib_d:7.1f} own {own_d:7.1f} | wgrad lib {lib_w:7.1f} own {own_w:7.1f}  (us, both mixers)", flush=True)
