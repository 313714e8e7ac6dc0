"""Developer tool: do independent branches of a captured hipGraph overlap on the device?  Two chains of small launches,
captured serially on one stream vs forked onto two streams; prints replay times."""
import torch
dev = torch.device("cuda")
a = [torch.randn(300, 300, device=dev) for _ in range(2)]
w = [torch.randn(300, 300, device=dev) * 0.05 for _ in range(2)]


def chain(i, n=20):
    x = a[i]
    for _ in range(n):
        x = torch.tanh(x @ w[i])
    return x


def timed(g):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(0); chain(1)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s):
    y0 = chain(0)
    y1 = chain(1)
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g2, stream=s):
    side.wait_stream(s)
    with torch.cuda.stream(side):
        z1 = chain(1)
    z0 = chain(0)
    s.wait_stream(side)
print("serial   %.1f us" % timed(g1))
print("forked   %.1f us" % timed(g2))
print("same results", torch.allclose(y0, z0), torch.allclose(y1, z1))
