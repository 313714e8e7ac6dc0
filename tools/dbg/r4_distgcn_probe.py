"""Debug probe: batch 1 of the S-FSQ parity fixture through the three-launch distance GCN and through the launch-per-product
path -- logits / loss / gradients against the oracle and against each other."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_bench_parity as T
from mobgt_amd import workloads

DEV = "cuda"
uni, model, coll = workloads.build("fsq", DEV, seed=1, model_overrides=dict(dropout_rate=0.0, intput_dropout_rate=0.0,
                                   attention_dropout_rate=0.0, warmup_updates=4, tot_updates=100, peak_lr=2e-3))
pool = workloads.make_pool("fsq", 2, 16, uni)
batches = [coll(t) for t in pool]
consts = T.oracle_consts(uni, model, "fsq")
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
model.eval()
for bi in (0, 1):
    b = batches[bi]
    ref_logits, ref_loss, ref_grads = T.oracle_step(sd0, T.cpu_batch(b), consts, 6)
    res = {}
    for tag, env in (("fused", "0"), ("split", "1"), ("fused2", "0")):
        os.environ["MOBGT_NO_DIST_GCN_FUSED"] = env
        for p in model.parameters():
            p.grad = None
        logits = model(b)[0]
        loss = model.training_step(b, 0)
        loss.backward()
        torch.cuda.synchronize()
        res[tag] = (logits.detach().float().cpu(), float(loss), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None})
    for tag in res:
        lg, ls, gr = res[tag]
        print("batch %d %-6s N %d: max|logit err| %.4e  loss %.7f (oracle %.7f)" % (bi, tag, b.x.shape[1], float((lg - ref_logits).abs().max()), ls, ref_loss))
    print("   logits fused-split max %.3e   fused-fused2 max %.3e" % (float((res["fused"][0] - res["split"][0]).abs().max()), float((res["fused"][0] - res["fused2"][0]).abs().max())))
    rel = lambda a, c: float((a - c).norm() / (c.norm() + 1e-30))
    for n in T.GRAD_PARAMS:
        r = ref_grads[n]
        print("   %-46s vs oracle: fused %.4f split %.4f fused2 %.4f | fused vs split %.4f  fused vs fused2 %.4f" % (
            n, rel(res["fused"][2][n], r), rel(res["split"][2][n], r), rel(res["fused2"][2][n], r),
            rel(res["fused"][2][n], res["split"][2][n]), rel(res["fused"][2][n], res["fused2"][2][n])))
