"""What a kernel node of a replayed hipGraph costs on this box (dev): a chain of N dependent tiny kernels, eager on one stream vs one graph replay,
and the same with the chain split over two forked streams (independent halves)."""
import torch
x = torch.zeros(64, device="cuda")
y = torch.zeros(64, device="cuda")
N = 400


def chain(t, n):
    for _ in range(n):
        t.add_(1.0)


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(x, 10)
torch.cuda.synchronize()
print("eager, one stream: %.2f us per kernel (host-bound if > graph)" % (timeit(lambda: chain(x, N)) / N))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    chain(x, N)
print("graph, linear chain of %d: %.2f us per kernel node" % (N, timeit(g.replay) / N))
g2 = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
with torch.cuda.graph(g2, stream=s):
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        chain(y, N // 2)
    chain(x, N // 2)
    torch.cuda.current_stream().wait_stream(s2)
print("graph, two independent chains of %d: %.2f us per kernel node" % (N // 2, timeit(g2.replay) / N))
big = torch.zeros(1 << 22, device="cuda")
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=s):
    for _ in range(100):
        big.add_(1.0)
t = timeit(g3.replay) / 100
print("graph, chain of 16 MB element-wise kernels: %.2f us each (%.2f TB/s)" % (t, 2 * big.numel() * 4 / t / 1e6))
