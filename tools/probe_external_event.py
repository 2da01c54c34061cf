"""Probe (MI355X): can a hipGraph replay release work on ANOTHER stream before the graph has finished?  An event created with
external=True and recorded DURING capture becomes an event-record node; a side stream that waits on it after graph.replay() should
start as soon as the node's predecessors are done.  Prints the wall times of: graph alone, side work alone, both serialised, both
with the external event.  (The graphed data-parallel step wants this: all-reduce group g while the backward of the earlier blocks
still runs.)"""
import time
import torch

dev = torch.device("cuda", 0)
x = torch.randn(4096, 4096, device=dev)
y = torch.randn(4096, 4096, device=dev)
out_a = torch.empty_like(x)
out_b = torch.empty_like(x)
z = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream(device=dev)


def busy(a, b, o, n):
    for _ in range(n):
        torch.mm(a, b, out=o)


def wall(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for external in (True,):
    ev = torch.cuda.Event(external=external)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        busy(x, y, out_a, 2)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        busy(x, y, out_a, 2)          # "the last blocks' backward": its result is what the side stream may touch
        ev.record()
        busy(x, y, out_b, 40)         # "the rest of the backward"
    torch.cuda.synchronize()

    def graph_only():
        g.replay()

    def side_only():
        with torch.cuda.stream(side):
            z.mul_(1.0001)
            for _ in range(30):
                z.add_(1e-6)
        torch.cuda.current_stream().wait_stream(side)

    def serial():
        g.replay()
        side.wait_stream(torch.cuda.current_stream())
        side_only()

    def overlapped():
        g.replay()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            z.mul_(1.0001)
            for _ in range(30):
                z.add_(1e-6)
        torch.cuda.current_stream().wait_stream(side)

    print(f"external={external}: graph {wall(graph_only):.2f} ms, side {wall(side_only):.2f} ms, serial {wall(serial):.2f} ms, "
          f"with event {wall(overlapped):.2f} ms", flush=True)
