"""Developer tool (GPU box): the scatter launch of the encoder input's backward (mobgt_embed_gather_multi, backward) at the S-FSQ
shape, all jobs and subsets -- which tables' atomics cost what."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import _lib
from mobgt_amd.ops import _stream
dev = "cuda"
G, N = 16, 38
R = G * N
g = torch.Generator(device="cpu").manual_seed(0)
jobs = {
    "poi":    (torch.zeros(R, 128, device=dev), torch.arange(R, device=dev), 0, 0, 128, -1),
    "time":   (torch.zeros(49, 32, device=dev), torch.randint(1, 49, (R,), generator=g).to(dev), 0, 128, 32, 0),
    "cat":    (torch.zeros(300, 32, device=dev), torch.randint(0, 300, (R,), generator=g).to(dev), 1, 160, 32, -1),
    "indeg":  (torch.zeros(128, 192, device=dev), torch.randint(1, 5, (R,), generator=g).to(dev), 2, 0, 192, 0),
    "outdeg": (torch.zeros(128, 192, device=dev), torch.randint(1, 5, (R,), generator=g).to(dev), 2, 0, 192, 0),
    "pe":     (torch.zeros(2000, 192, device=dev), (torch.arange(R, device=dev) % N) + 1, 2, 0, 192, -1),
}
bufs = [torch.randn(R, 160, device=dev), torch.randn(R, 192, device=dev), torch.randn(R, 192, device=dev)]
vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


def call(names):
    m = len(names)
    js = [jobs[n] for n in names]
    _lib.check(_lib.lib().mobgt_embed_gather_multi(
        m, None, (vp * m)(*[j[0].data_ptr() for j in js]), (vp * m)(*[j[1].data_ptr() for j in js]),
        (i64 * m)(*[j[5] for j in js]), (ci * m)(*[j[4] for j in js]), (ci * m)(*[j[3] for j in js]), None,
        (vp * m)(*[bufs[j[2]].data_ptr() for j in js]), (i64 * m)(*[bufs[j[2]].stride(0) for j in js]), R, 0, 1, None, 0, _stream()), "bwd")


def timeit(f, n=30):
    filler = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, stream=s):
            for i in range(n):
                filler.fill_(i); f()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            for i in range(n):
                filler.fill_(i)
    torch.cuda.synchronize()
    out = []
    for q in (gr, g2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); q.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return out[0] - out[1]

allj = list(jobs)
print("all six: %.1f us" % timeit(lambda: call(allj)))
for n_ in allj:
    print("only %-7s %.1f us   without it %.1f us" % (n_, timeit(lambda: call([n_])), timeit(lambda: call([x for x in allj if x != n_]))))
print("without both degree tables: %.1f us" % timeit(lambda: call(["poi", "time", "cat", "pe"])))

# (round 4: a variant that gave the small tables workgroups of their own -- 64 positions each, the rows with index < 16 summed in
#  LDS or in registers, one atomic per (index, column) and workgroup -- measured 21-35 us for the launch against 12.8: a few
#  workgroups walking 64 rows each are a longer chain than 152 waves with one row each; not kept.)
