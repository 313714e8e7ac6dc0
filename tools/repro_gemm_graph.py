"""Is a library GEMM inside a replayed hipGraph sensitive to what later kernels of the same graph leave in
recycled pool memory (e.g. a split-K workspace that is expected to stay zero)?"""
import torch
dev = "cuda"
torch.manual_seed(0)
P = 7856
A = (torch.rand(P, P, device=dev) * 0.01).to(torch.bfloat16)
x = torch.randn(P, 16, device=dev)
W = torch.randn(16, 64, device=dev) * 0.1
outs = {}
def body():
    s = (x @ W).to(torch.bfloat16)
    o = torch.mm(A, s).float()
    # poison: allocate-and-free buffers of many sizes filled with NaN so that recycled pool memory is dirty
    for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        t = torch.full((n,), float("nan"), device=dev)
        del t
    return o
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    ref = body()
torch.cuda.current_stream().wait_stream(st)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st):
    out = body()
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "nan:", torch.isnan(out).any().item(), "maxdiff vs eager:", (out - ref).abs().max().item())
