"""mobgt_layer_wgrad_big alone at S-BIG sizes: 12 operand sets (1.2 GB: every launch reads cold HBM), 20 rounds, HIP events around
graph-free launches on the current stream; against the library path it replaces (three split-K bmm + sums + one mm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops
R, C, F = (int(v) for v in (sys.argv[1:4] + [12560, 256, 1024][len(sys.argv) - 1:]))
dev = "cuda"
bf = dict(dtype=torch.bfloat16, device=dev)
sets = []
for i in range(12):
    sets.append([(torch.randn(R, C, **bf), torch.randn(R, F, **bf), None, None), (torch.randn(R, F, **bf), torch.randn(R, C, **bf), None, None),
                 (torch.randn(R, C, **bf), torch.randn(R, C, **bf), None, None),
                 (torch.randn(R, 3 * C, **bf), torch.randn(R, C, **bf), torch.zeros(3 * C, device=dev), None)])
def run_big():
    for it in sets:
        ops.layer_wgrad_big(it, R)
def run_lib():
    from mobgt_amd.fused_layer import _mm_tn_f32
    for it in sets:
        for g, x, _, _ in it:
            _mm_tn_f32(g, x)
for name, fn in (("wgrad_big (+ 4 part.sum(0))", run_big), ("library split-K", run_lib)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-30s %.1f us per layer" % (name, e0.elapsed_time(e1) * 1000 / 120))
