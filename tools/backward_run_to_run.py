"""Developer tool (GPU box): run-to-run spread of the long-batch step's gradients (f32 atomics + bf16 rounding points), per form of the step,
hipGraph and eager: profiles/r6_backward_run_to_run.txt.  Dropout off, same model, same batch, 8 steps-from-scratch per mode."""
import os, sys, gc
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from mobgt_amd import fused_layer, ops
from mobgt_amd.train import TrainStep
from test_gpu_long_parity import _build
def run(w, b, graph=True):
    fused_layer._WGRAD_BIG[0], ops._BIAS_BWD_BESIDE[0] = w, b
    uni, model, batch = _build(192, dropout_rate=0.0, intput_dropout_rate=0.0, attention_dropout_rate=0.0)
    ts = TrainStep(model, [batch], use_graph=graph, seed=5)
    ts.prepare(); loss = float(ts.step(0)); torch.cuda.synchronize()
    g = {n: q.grad.detach().double().clone() for n, q in model.named_parameters() if q.grad is not None}
    del ts, model; gc.collect()
    return loss, g
def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-300))
_, ref = run(True, True)
names = [n for n in ref if not n.endswith("linear_k.bias")]
for mode in [(True, True, True), (False, False, True), (True, True, False), (False, False, False)]:
    for i in range(8):
        loss, g = run(*mode)
        d = sorted(((rel(g[n], ref[n]), n) for n in names), reverse=True)
        nz = sum(1 for r, _ in d if r > 5e-5)
        print(mode, i, "loss %.9f" % loss, "n>5e-5: %3d" % nz, " ".join("%s %.1e" % (n[-28:], r) for r, n in d[:3]), flush=True)
