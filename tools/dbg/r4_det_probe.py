"""Probe (GPU box): how far do the bias / edge / degree table gradients of the dropout-ON S-FSQ train step move towards the
oracle when the dBias path keeps f32 (bias dtype f32: one f32 accumulator instead of bf16 slices)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from mobgt_amd import workloads
from mobgt_amd.train import TrainStep
from oracle import model_oracle as mo
from test_gpu_bench_parity import LOSS_SCALE, check_grad, cpu_batch, oracle_consts
from test_gpu_train_parity import step_masks, _drop_hook
TABLES = ["rel_pos_encoder.weight", "poi_pos_encoder.weight", "edge_encoder.weight", "edge_dis_encoder.weight",
          "graph_token_virtual_distance.weight", "in_degree_encoder.weight", "pos_embed.pe", "time_embed_model_48.weight"]
for dtype in ("bf16", "f32"):
    uni, model, coll = workloads.build("fsq", "cuda", seed=1, dtype=dtype, gemm_dtype="bf16",
                                       model_overrides=dict(warmup_updates=4, tot_updates=100, peak_lr=2e-3))
    batches = [coll(t) for t in workloads.make_pool("fsq", 2, 16, uni)]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    consts = oracle_consts(uni, model, "fsq")
    ts = TrainStep(model, batches, use_graph=True, seed=5)
    ts.prepare()
    params = dict(model.named_parameters())
    for i, b in enumerate(batches):
        with torch.no_grad():
            model.load_state_dict(sd0); ts.sync_shadows()
        loss = float(ts.step(i))
        step = int(ts.seed_dev.item())
        masks = step_masks(model, b, step, 5)
        sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
        ref_loss = mo.fq_training_loss(sd, cpu_batch(b), consts, n_layers=6, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True,
                                       hidden=model.hidden_dim, drop=_drop_hook(masks))
        (ref_loss * LOSS_SCALE).backward()
        rep = []
        for n in TABLES:
            check_grad(n, params[n].grad, sd[n].grad / LOSS_SCALE, rep)
        print(dtype, "batch", i, "loss", loss, float(ref_loss), {r[0].split(".")[0]: round(r[2], 4) for r in rep})
    del ts, model
