"""Developer tool: dump HIP vs oracle gradients of the benched S-FSQ configuration to gpurun_out/parity_dump.npz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mobgt_amd import workloads
from oracle import model_oracle as mo
import test_gpu_bench_parity as T

uni, model, coll = workloads.build("fsq", "cuda", seed=1, model_overrides=dict(dropout_rate=0.0, intput_dropout_rate=0.0, attention_dropout_rate=0.0))
b = coll(workloads.make_pool("fsq", 1, 16, uni)[0])
consts = T.oracle_consts(uni, model, "fsq")
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
out = {}
for scale in (1.0, 65536.0):
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
    logits, _ = mo.graphormer_fq_forward(sd, T.cpu_batch(b), consts, n_layers=6, H=8, D=20)
    loss = mo.gradient_tail_loss(logits, T.cpu_batch(b).y - 1, 0.2)
    (loss * scale).backward()
    for n in T.GRAD_PARAMS:
        if n != "out_proj.weight":
            out[f"ref{int(scale)}/{n}"] = (sd[n].grad / scale).numpy()
model.eval()
model.training_step(b, 0).backward()
for n, p in model.named_parameters():
    if n in T.GRAD_PARAMS and n != "out_proj.weight":
        out[f"got/{n}"] = p.grad.float().cpu().numpy()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "parity_dump.npz"), **out)
print("ok")
